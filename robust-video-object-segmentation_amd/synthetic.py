"""Seeded synthetic feature-level clips.

No datasets or checkpoints exist in the build environment, so benchmarks and parity tests run
on synthetic *feature maps*: stride-4 embeddings shaped like the output of the reference's
``semantic_embedding`` head (ReLU-terminated, aocnet.py:19-25) and integer label maps with
moving blobs over background id 0.  numpy only (RandomState is platform independent), so the
same seed yields the same clip in the build container and on the GPU box.
"""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class ClipConfig:
    """One of BASELINE.json's configs at feature level (SURVEY.md section 8 sizes)."""
    name: str
    h: int
    w: int
    n_obj: int          # O: objects INCLUDING background (gt_ids + 1, aocnet.py:141-142)
    k: int = 16         # proxies per object (AEM:232)
    c: int = 100        # MODEL_SEMANTIC_EMBEDDING_DIM (configs/resnet101_aocnet.py:65)
    frames: int = 60
    mem_every: int = 5  # MEM_EVERY (configs/resnet101_aocnet.py:17)


CONFIGS = {
    # 481x849 input -> 121x213 stride-4 map (custom_transforms.py:427-430, resnet.py:109-115)
    "cfg1": ClipConfig("cfg1", 121, 213, 2, 16, frames=2),
    "cfg2": ClipConfig("cfg2", 121, 213, 4, 16, frames=60),
    "cfg3": ClipConfig("cfg3", 145, 261, 6, 16, frames=30),
    "cfg4": ClipConfig("cfg4", 181, 321, 9, 64, frames=8),
    # small maps for CPU-side tests
    "tiny": ClipConfig("tiny", 24, 40, 3, 16, frames=6),
}


def _box3(x):
    """3x3 box filter with edge replication over the first two axes."""
    p = np.pad(x, ((1, 1), (1, 1), (0, 0)), mode="edge")
    out = np.zeros_like(x)
    for dy in range(3):
        for dx in range(3):
            out += p[dy:dy + x.shape[0], dx:dx + x.shape[1]]
    return out / 9.0


def fresh_embedding(rng, h, w, c):
    """relu(N(0,1)) * 0.3, box-smoothed: keeps squared distances inside the informative range
    of the reference's (sigmoid(d + b) - 0.5) * 2 transform."""
    x = np.maximum(rng.randn(h, w, c), 0.0).astype(np.float32) * np.float32(0.3)
    return _box3(x).astype(np.float32)


def next_embedding(rng, prev, h, w, c):
    """Temporal coherence: E_t = 0.9 E_{t-1} + 0.1 fresh."""
    return (np.float32(0.9) * prev + np.float32(0.1) * fresh_embedding(rng, h, w, c)).astype(np.float32)


def blob_tracks(rng, h, w, n_obj):
    """Per foreground object: centre, half sizes, velocity, shape (0 = box, 1 = ellipse)."""
    tracks = []
    for _ in range(1, n_obj):
        hh = max(2, int(h * rng.uniform(0.10, 0.18)))
        hw = max(2, int(w * rng.uniform(0.08, 0.14)))
        cy = rng.uniform(hh, h - hh)
        cx = rng.uniform(hw, w - hw)
        vy, vx = rng.uniform(-0.6, 0.6), rng.uniform(-1.0, 1.0)
        tracks.append([cy, cx, hh, hw, vy, vx, int(rng.randint(0, 2))])
    return tracks


def label_map(tracks, t, h, w):
    """Integer label map [h, w] at time t: id 0 background, object j at id j (later ids on top)."""
    yy, xx = np.mgrid[0:h, 0:w]
    lab = np.zeros((h, w), np.int32)
    for j, (cy, cx, hh, hw, vy, vx, shape) in enumerate(tracks, start=1):
        y = (cy + vy * t - hh) % max(1, h - 2 * hh) + hh if h > 2 * hh else cy
        x = (cx + vx * t - hw) % max(1, w - 2 * hw) + hw if w > 2 * hw else cx
        if shape == 0:
            m = (np.abs(yy - y) <= hh) & (np.abs(xx - x) <= hw)
        else:
            m = ((yy - y) / hh) ** 2 + ((xx - x) / hw) ** 2 <= 1.0
        lab[m] = j
    return lab


def one_hot(lab, n_obj):
    """[h, w] int -> [h, w, O] float32 0/1 (aocnet.py:154,191: (label == ids).float())."""
    return (lab[..., None] == np.arange(n_obj, dtype=np.int32)[None, None, :]).astype(np.float32)


def make_clip(cfg, seed=0, frames=None):
    """Returns dict(emb=[T,h,w,C] float32, lab=[T,h,w] int32) for one synthetic sequence."""
    rng = np.random.RandomState(seed)
    frames = cfg.frames if frames is None else frames
    tracks = blob_tracks(rng, cfg.h, cfg.w, cfg.n_obj)
    embs, labs = [], []
    e = fresh_embedding(rng, cfg.h, cfg.w, cfg.c)
    for t in range(frames):
        if t > 0:
            e = next_embedding(rng, e, cfg.h, cfg.w, cfg.c)
        embs.append(e)
        labs.append(label_map(tracks, t, cfg.h, cfg.w))
    return dict(emb=np.stack(embs), lab=np.stack(labs))


def kmeans_init_rows(seed, counts, k):
    """Explicit k-means initial rows per object, drawn as scipy's minit='points' does on a
    RandomState (``permutation(n_i)[:K_i]``) with the reference's sticky K (AEM:268)."""
    rng = np.random.RandomState(seed)
    rows, kk = [], k
    for n_i in counts:
        kk = min(kk, int(n_i))
        rows.append(rng.permutation(int(n_i))[:kk].astype(np.int64) if kk > 0 else None)
    return rows
