// Host check of maggie_amd/csrc/loss_stencils.h (the per-cell arithmetic of the loss stencils' batched-load kernels, csrc/losses.hip):
//   1. ring_map enumerates every cell of the grid exactly once, and flags as "inner" exactly the cells outside the ring;
//   2. on every inner cell the batched form equals the general walk (the kernels take the general walk everywhere else);
//   3. the general forms (loads batched, conditions applied when the terms are added) equal the tap-by-tap walks they replaced on every cell, exactly.
// Built and run by tests/test_loss_stencils_cpu.py with g++ (no GPU, no HIP): exit code 0 = all sizes agree.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define MG_STENCIL_FN static inline
#define __restrict__
#include "../../maggie_amd/csrc/loss_stencils.h"

// ---- the tap-by-tap walks the batched general forms replaced (one load behind its condition per tap): the reference for every cell, bit for bit ----
static float ref_lap_up(const float* dp, int y, int xx, int h, int w, int wd) {
    const float g[5] = MG_G1;
    float up = 0.f;
    for (int i = 0; i < 5; ++i) {
        int yy = st_refl(y + i - 2, h);
        if (yy & 1) continue;
        float r = 0.f;
        for (int j = 0; j < 5; ++j) {
            int xs = st_refl(xx + j - 2, w);
            if (!(xs & 1)) r += g[j] * dp[(yy >> 1) * wd + (xs >> 1)];
        }
        up += g[i] * r;
    }
    return up;
}
static float ref_upT(const float* qp, int a, int b, int h, int w) {
    const float g[5] = MG_G1;
    float acc = 0.f;
    for (int sa = 0; sa < 3; ++sa) {
        int ma = sa == 0 ? 2 * a : (sa == 1 ? -2 * a : 2 * (h - 1) - 2 * a);
        if (sa == 1 && a != 1) continue;
        if (sa == 2 && ma != h) continue;
        for (int i = 0; i < 5; ++i) {
            int y = ma - i + 2;
            if (y < 0 || y >= h) continue;
            float rowacc = 0.f;
            for (int sb = 0; sb < 3; ++sb) {
                int mb = sb == 0 ? 2 * b : (sb == 1 ? -2 * b : 2 * (w - 1) - 2 * b);
                if (sb == 1 && b != 1) continue;
                if (sb == 2 && mb != w) continue;
                for (int j = 0; j < 5; ++j) {
                    int xq = mb - j + 2;
                    if (xq < 0 || xq >= w) continue;
                    rowacc += g[j] * qp[y * w + xq];
                }
            }
            acc += g[i] * rowacc;
        }
    }
    return acc;
}
static float ref_downT(const float* rp, int Y, int X, int h, int w, int hd, int wd) {
    const float g[5] = MG_G1;
    float acc = 0.f;
    for (int sa = 0; sa < 3; ++sa) {
        int my = sa == 0 ? Y : (sa == 1 ? -Y : 2 * (h - 1) - Y);
        if (sa == 1 && !(Y == 1 || Y == 2)) continue;
        if (sa == 2 && !(Y == h - 2 || Y == h - 3)) continue;
        for (int i = 0; i < 5; ++i) {
            int ty = my - i + 2;
            if (ty < 0 || (ty & 1)) continue;
            int y = ty >> 1;
            if (y >= hd) continue;
            float rowacc = 0.f;
            for (int sb = 0; sb < 3; ++sb) {
                int mx = sb == 0 ? X : (sb == 1 ? -X : 2 * (w - 1) - X);
                if (sb == 1 && !(X == 1 || X == 2)) continue;
                if (sb == 2 && !(X == w - 2 || X == w - 3)) continue;
                for (int j = 0; j < 5; ++j) {
                    int tx = mx - j + 2;
                    if (tx < 0 || (tx & 1)) continue;
                    int xr = tx >> 1;
                    if (xr >= wd) continue;
                    rowacc += g[j] * rp[y * wd + xr];
                }
            }
            acc += g[i] * rowacc;
        }
    }
    return acc;
}

static int fails = 0;
static void expect(bool ok, const char* what, int h, int w, int a, int b) {
    if (!ok && fails++ < 20) std::fprintf(stderr, "FAIL %s at plane %dx%d cell (%d, %d)\n", what, h, w, a, b);
}
static bool close(float x, float y) { return std::fabs(x - y) <= 1e-6f * (1.f + std::fabs(x) + std::fabs(y)); }

static void check_ring(int nh, int nw, int T, int Bt, int Lw, int Rw) {
    std::vector<int> seen((size_t)nh * nw, 0);
    const bool any_inner = nh - T - Bt > 0 && nw - Lw - Rw > 0;
    bool border_started = false;
    for (int t = 0; t < nh * nw; ++t) {
        int a = -1, b = -1;
        const bool in = ring_map(t, nh, nw, T, Bt, Lw, Rw, a, b);
        expect(a >= 0 && a < nh && b >= 0 && b < nw, "ring_map range", nh, nw, a, b);
        if (a < 0 || a >= nh || b < 0 || b >= nw) continue;
        seen[(size_t)a * nw + b]++;
        const bool want = any_inner && a >= T && a < nh - Bt && b >= Lw && b < nw - Rw;
        expect(in == want, "ring_map inner flag", nh, nw, a, b);
        if (!in) border_started = true;
        expect(!(in && border_started), "inner cells come first", nh, nw, a, b);
    }
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) expect(seen[(size_t)a * nw + b] == 1, "ring_map bijection", nh, nw, a, b);
}

static void check_plane(int h, int w, unsigned seed) {
    const int hd = h / 2, wd = w / 2;
    std::srand(seed);
    auto rnd = [] { return (float)std::rand() / (float)RAND_MAX - 0.5f; };
    std::vector<float> full((size_t)h * w), half((size_t)hd * wd), A((size_t)h * w), B((size_t)h * w);
    for (auto& v : full) v = rnd();
    for (auto& v : half) v = rnd();
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    check_ring(hd, wd, 1, 1, 1, 1);
    check_ring(hd, wd, 2, 1, 2, 1);
    check_ring(hd, wd, 2, 2, 2, 2);
    check_ring(h, w, 1, 1, 1, 1);
    for (int t = 0; t < hd * wd; ++t) {
        int a, b;
        if (ring_map(t, hd, wd, 1, 1, 1, 1, a, b)) {                   // pyr_lap_fwd quads
            float up[4];
            lap_up_inner(half.data(), a, b, wd, up);
            for (int u = 0; u < 4; ++u)
                expect(close(up[u], lap_up_general(half.data(), 2 * a + (u >> 1), 2 * b + (u & 1), h, w, wd)), "lap_up", h, w, 2 * a + (u >> 1), 2 * b + (u & 1));
        }
        if (ring_map(t, hd, wd, 2, 1, 2, 1, a, b))                     // pyr_upT cells
            expect(close(upT_inner(full.data(), a, b, w), upT_general(full.data(), a, b, h, w)), "upT", h, w, a, b);
        if (ring_map(t, hd, wd, 2, 2, 2, 2, a, b)) {                   // pyr_downT quads
            float acc[4];
            downT_inner(half.data(), a, b, wd, acc);
            for (int u = 0; u < 4; ++u)
                expect(close(acc[u], downT_general(half.data(), 2 * a + (u >> 1), 2 * b + (u & 1), h, w, hd, wd)), "downT", h, w, 2 * a + (u >> 1), 2 * b + (u & 1));
        }
    }
    for (int Y = 0; Y < h; ++Y)                                          // the general forms against the tap-by-tap walks: every cell, exact
        for (int X = 0; X < w; ++X) {
            expect(lap_up_general(half.data(), Y, X, h, w, wd) == ref_lap_up(half.data(), Y, X, h, w, wd), "lap_up_general vs walk", h, w, Y, X);
            expect(downT_general(half.data(), Y, X, h, w, hd, wd) == ref_downT(half.data(), Y, X, h, w, hd, wd), "downT_general vs walk", h, w, Y, X);
        }
    for (int a = 0; a < hd; ++a)
        for (int b = 0; b < wd; ++b) expect(upT_general(full.data(), a, b, h, w) == ref_upT(full.data(), a, b, h, w), "upT_general vs walk", h, w, a, b);
    for (int t = 0; t < h * w; ++t) {
        int Y, X;
        if (ring_map(t, h, w, 1, 1, 1, 1, Y, X))                       // point_bwd pixels
            expect(close(sobel_adj_inner(A.data(), B.data(), Y, X, w), sobel_adj_general(A.data(), B.data(), Y, X, h, w)), "sobel_adj", h, w, Y, X);
    }
}

int main() {
    const int sizes[][2] = {{4, 4}, {4, 6}, {6, 4}, {8, 8}, {10, 18}, {16, 16}, {18, 10}, {20, 36}, {32, 32}, {40, 72}, {64, 64}, {128, 96}};
    unsigned seed = 1;
    for (const auto& s : sizes) check_plane(s[0], s[1], seed++);
    if (fails) { std::fprintf(stderr, "%d mismatches\n", fails); return 1; }
    std::puts("loss stencils: batched forms agree with the general walks on every inner cell; ring_map is a bijection");
    return 0;
}
