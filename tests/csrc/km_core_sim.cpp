// Host replay of the chunk-parallel exact sums (robust-video-object-segmentation_amd/csrc/km_exact_core.h): random chains are cut into
// chunks, folded with predicted prefixes, merged into runs and stitched exactly as kmeans_persistent.hip does it per lane, and the
// result is compared bit for bit with the literal sequential float32 sum.  Test infrastructure only (tests/test_km_exact_core.py).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../robust-video-object-segmentation_amd/csrc/km_exact_core.h"

struct Rec {
    uint32_t hdr;
    int32_t A0, B0;
    float lits[KX_MAX_LIT];
    uint32_t run_hdr;       // run in front of this record (phase C)
    int32_t run_R0;
};

struct Stats {
    long chunks = 0, plain = 0, cross = 0, set = 0, unsafe = 0, rec_fail = 0, run_fail = 0, lits = 0, events = 0;
};

static float literal_sum(const std::vector<float> &x, size_t a, size_t b, float s) {
    for (size_t i = a; i < b; ++i) s = s + x[i];
    return s;
}

// one chain: members x, chunk boundaries cb (cb[c] .. cb[c+1]), prediction noise (relative), part size in chunks
static bool run_chain(const std::vector<float> &x, const std::vector<size_t> &cb, double noise, int part, std::mt19937 &rng, Stats &st,
                      bool verbose) {
    const size_t nc = cb.size() - 1;
    std::vector<Rec> rec(nc);
    // ---- phase A: any-order chunk sums of |x| (here: reversed order), sequential float prefix over the chunks
    std::vector<float> P(nc);
    std::vector<int> mb(nc);
    float run = 0.0f;
    std::uniform_real_distribution<double> un(-1.0, 1.0);
    for (size_t c = 0; c < nc; ++c) {
        P[c] = run;
        if (noise > 0 && run > 0) P[c] = (float)(run * (1.0 + noise * un(rng)));
        mb[c] = (int)cb[c];
        float bs = 0.0f;
        for (size_t i = cb[c + 1]; i > cb[c]; --i) bs = bs + std::fabs(x[i - 1]);
        run = run + bs;
    }
    // ---- phase B: fold
    for (size_t c = 0; c < nc; ++c) {
        KxFold k;
        kx_fold_init(k, P[c], mb[c]);
        for (size_t i = cb[c]; i < cb[c + 1]; ++i) {
            const KxFast f = kx_fold_fast(k, x[i]);
            if (f.over || (rng() % 16 == 0)) kx_fold_step(k, f, x[i], (int)(i - cb[c]), rec[c].lits, 1);      // a wave takes the general step when ANY lane needs it
            else { k.acc = f.acc; k.dvar = f.dvar; k.s = k.s + x[i]; }
        }
        rec[c].hdr = kx_fold_finish(k, (int)(cb[c + 1] - cb[c]), rec[c].A0, rec[c].B0);
        rec[c].run_hdr = 0; rec[c].run_R0 = 0;
        ++st.chunks;
        switch (kx_hdr_kind(rec[c].hdr)) {
        case KX_PLAIN: ++st.plain; break;
        case KX_CROSS: ++st.cross; st.lits += kx_hdr_nlit(rec[c].hdr); break;
        case KX_SET: ++st.set; break;
        default: ++st.unsafe;
        }
    }
    // ---- phase C per part: runs of PLAIN records; non-plain records carry the run in front of them, the part its last run
    const size_t np_parts = (nc + part - 1) / part;
    std::vector<KxRun> post(np_parts);
    std::vector<std::vector<size_t>> nonplain(np_parts);
    for (size_t p = 0; p < np_parts; ++p) {
        KxRun r{0, 0, 0};
        for (size_t c = p * part; c < std::min(nc, (p + 1) * part); ++c) {
            if (kx_run_merge(r, rec[c].hdr, rec[c].A0)) continue;
            rec[c].run_hdr = kx_run_hdr(r);
            rec[c].run_R0 = r.R0;
            nonplain[p].push_back(c);
            r = KxRun{0, 0, 0};
        }
        post[p] = r;
    }
    // ---- phase S: exact state through the parts
    float s = 0.0f;
    for (size_t p = 0; p < np_parts; ++p) {
        const size_t c_end = std::min(nc, (p + 1) * part);
        size_t done = p * part;             // chunks [p * part, done) are in the state
        bool expanded = false;
        auto chunk_by_chunk = [&](size_t from, size_t to) {
            for (size_t c = from; c < to; ++c) {
                if (!kx_apply_record(s, rec[c].hdr, rec[c].A0, rec[c].B0, rec[c].lits)) {
                    ++st.rec_fail;
                    s = literal_sum(x, cb[c], cb[c + 1], s);
                }
            }
        };
        for (size_t q = 0; q < nonplain[p].size() && !expanded; ++q) {
            const size_t c = nonplain[p][q];
            ++st.events;
            const KxRun r = kx_run_unpack(rec[c].run_hdr, rec[c].run_R0);
            float t = s;
            if (!kx_apply_run(t, r)) {      // some chunk of the run did not happen as predicted: walk the rest of the part chunk by chunk
                ++st.run_fail;
                chunk_by_chunk(done, c_end);
                expanded = true;
                break;
            }
            s = t;
            chunk_by_chunk(c, c + 1);
            done = c + 1;
        }
        if (!expanded) {
            float t = s;
            ++st.events;
            if (!kx_apply_run(t, post[p])) { ++st.run_fail; chunk_by_chunk(done, c_end); }
            else s = t;
        }
    }
    const float want = literal_sum(x, 0, x.size(), 0.0f);
    const bool ok = kx_f2u(want) == kx_f2u(s) || (want != want && s != s);
    if (!ok && verbose) std::printf("MISMATCH: want %.9g (%08x) got %.9g (%08x), %zu members %zu chunks\n", want, kx_f2u(want), s, kx_f2u(s), x.size(), nc);
    return ok;
}

int main(int argc, char **argv) {
    const int trials = argc > 1 ? std::atoi(argv[1]) : 2000;
    const unsigned seed = argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u;
    std::mt19937 rng(seed);
    Stats total;
    int bad = 0;
    const char *names[] = {"relu*0.3", "uniform", "few-bits (ties)", "zeros then values", "wide range", "signed", "tiny+big", "constant", "nan/inf"};
    for (int t = 0; t < trials; ++t) {
        const int kind = t % 9;
        std::uniform_int_distribution<int> len_d(1, kind == 4 ? 3000 : 60000);
        size_t n = (size_t)len_d(rng);
        if (t % 17 == 0) n = 1 + n % 70;
        if (t % 29 == 0) n = 150000 + n * 5;
        std::vector<float> x(n);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::uniform_real_distribution<float> ud(0.f, 1.f);
        for (size_t i = 0; i < n; ++i) {
            float v;
            switch (kind) {
            case 0: v = std::fmax(nd(rng), 0.f) * 0.3f; break;
            case 1: v = ud(rng); break;
            case 2: v = (float)(rng() % 64) * 0.015625f; break;              // multiples of 2^-6: ties everywhere
            case 3: v = i < n / 2 ? 0.f : std::fmax(nd(rng), 0.f); break;
            case 4: v = std::exp(nd(rng) * 6.f); break;
            case 5: v = nd(rng); break;
            case 6: v = (rng() % 50 == 0) ? 1000.f * ud(rng) : 1e-4f * ud(rng); break;
            case 7: v = 0.1f; break;
            default: v = (i == n / 3) ? NAN : ((i == n / 2 && t % 2) ? INFINITY : ud(rng));
            }
            x[i] = v;
        }
        std::vector<size_t> cb{0};
        const int cmax = (t % 5 == 0) ? 3 : 64;
        std::uniform_int_distribution<int> cd(1, cmax);
        while (cb.back() < n) cb.push_back(std::min(n, cb.back() + (size_t)cd(rng)));
        const double noise = (t % 7 == 3) ? 1e-3 : (t % 7 == 5) ? 3e-6 : 0.0;
        const int part = (t % 3 == 0) ? 8 : 64;
        Stats st;
        if (!run_chain(x, cb, noise, part, rng, st, true)) {
            ++bad;
            std::printf("  trial %d kind %s n %zu noise %g\n", t, names[kind], n, noise);
        }
        total.chunks += st.chunks; total.plain += st.plain; total.cross += st.cross; total.set += st.set; total.unsafe += st.unsafe;
        total.rec_fail += st.rec_fail; total.run_fail += st.run_fail; total.lits += st.lits; total.events += st.events;
        if (argc > 3 && kind == std::atoi(argv[3]) && t < 90)
            std::printf("kind %-18s n %7zu noise %-6g chunks %6ld plain %6ld cross %4ld (lits %4ld) set %3ld unsafe %4ld | rec_fail %4ld run_fail %3ld events %5ld\n",
                        names[kind], n, noise, st.chunks, st.plain, st.cross, st.lits, st.set, st.unsafe, st.rec_fail, st.run_fail, st.events);
    }
    std::printf("trials %d mismatches %d | chunks %ld plain %ld cross %ld (lits %ld) set %ld unsafe %ld | record failures %ld run failures %ld events %ld\n",
                trials, bad, total.chunks, total.plain, total.cross, total.lits, total.set, total.unsafe, total.rec_fail, total.run_fail, total.events);
    return bad ? 1 : 0;
}
