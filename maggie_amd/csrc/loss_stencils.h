// Per-cell arithmetic of the Laplacian-pyramid / Sobel-adjoint stencils of csrc/losses.hip (reference: maggie/network/loss.py:67-191, LapLoss /
// GradientLoss; the adjoints are ours). Every stencil exists twice:
//   *_general : the walk over all taps with the reflect / replicate border folded into run-time conditions -- valid for every cell. Written tap by
//               tap (load behind its condition) it compiles to one load + one full wait per tap; the forms here request a pixel's 25 taps
//               (lap_up), or a row's 15 column candidates (upT / downT), as one batch and apply the conditions when the terms are added;
//   *_inner   : the same terms in the same order for a cell whose taps do not touch the border, where the tap set depends on the cell's parity
//               alone: the window is loaded unconditionally as one batch, then combined.
// The includer defines MG_STENCIL_FN (`__device__ __forceinline__` in losses.hip; `static inline` in tests/csrc/loss_stencils_check.cpp, which
// checks on the host that both forms agree on every inner cell, that the general forms equal the tap-by-tap walks they replaced on EVERY cell, bit for bit,
// and that ring_map enumerates every cell once).
#pragma once

MG_STENCIL_FN int st_refl(int k, int n) { return k < 0 ? -k : (k >= n ? 2 * (n - 1) - k : k); }
MG_STENCIL_FN int st_clampi(int k, int n) { return k < 0 ? 0 : (k >= n ? n - 1 : k); }

#define MG_G1 {1.f / 16.f, 4.f / 16.f, 6.f / 16.f, 4.f / 16.f, 1.f / 16.f}

// Index t of an (nh x nw) grid -> cell (a, b), enumerated so that the cells away from the border come first, row by row, and the border ring (top
// T rows, bottom Bt rows, left Lw / right Rw columns) last: the waves of a launch are then all-inner or all-border but for one. (With the plain
// row-major order the first and last wave of every row hold a border cell and walk BOTH paths -- half of the waves at 256 cells per row.)
// Returns true for an inner cell.
MG_STENCIL_FN bool ring_map(int t, int nh, int nw, int T, int Bt, int Lw, int Rw, int& a, int& b) {
    const int hi = nh - T - Bt, wi = nw - Lw - Rw;
    if (hi <= 0 || wi <= 0) { a = t / nw; b = t - a * nw; return false; }
    const int n_int = hi * wi;
    if (t < n_int) { const int r = t / wi; a = T + r; b = Lw + (t - r * wi); return true; }
    int u = t - n_int;
    const int n_tb = (T + Bt) * nw;
    if (u < n_tb) { const int r = u / nw; a = r < T ? r : nh - Bt + (r - T); b = u - r * nw; return false; }
    u -= n_tb;
    const int side = Lw + Rw;
    const int r = u / side, k = u - r * side;
    a = T + r; b = k < Lw ? k : nw - Rw + (k - Lw);
    return false;
}

// ---- up = gauss5 * zero_stuff(down) at pixel (y, xx) of the (h x w) plane; down is (h/2 x w/2) = (. x wd) ------------------------------------
MG_STENCIL_FN float lap_up_general(const float* __restrict__ dp, int y, int xx, int h, int w, int wd) {
    // the 5 x 5 reflected taps all land on a cell of `down` ((yy >> 1, xs >> 1) is always inside): they are loaded as one batch, the odd (zero-stuffed)
    // rows / columns are skipped when the terms are added -- the terms and their order are those of the tap-by-tap walk (border waves used to spend one
    // memory round trip per tap, which set an ~11 us floor under every launch of the pyramid kernels whatever the level's size)
    const float g[5] = MG_G1;
    int yy[5], xs[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { yy[i] = st_refl(y + i - 2, h); xs[i] = st_refl(xx + i - 2, w); }
    float v[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) v[i][j] = dp[(yy[i] >> 1) * wd + (xs[j] >> 1)];
    float up = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (yy[i] & 1) continue;
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (!(xs[j] & 1)) r += g[j] * v[i][j];
        }
        up += g[i] * r;
    }
    return up;
}
// the quad (2a + dy, 2b + dx), inner ring (1, 1, 1, 1) of the (h/2 x w/2) quad grid: up[2 * dy + dx]
MG_STENCIL_FN void lap_up_inner(const float* __restrict__ dp, int a, int b, int wd, float* up) {
    const float* __restrict__ r0 = dp + (a - 1) * wd + (b - 1);
    float d[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) d[r][c] = r0[r * wd + c];
    const float g0 = 1.f / 16.f, g1 = 4.f / 16.f, g2 = 6.f / 16.f, g3 = 4.f / 16.f, g4 = 1.f / 16.f;
    // (even y, even x): rows a-1, a, a+1 under g0, g2, g4; columns b-1, b, b+1 under g0, g2, g4
    float u = 0.f;
    u += g0 * (g0 * d[0][0] + g2 * d[0][1] + g4 * d[0][2]);
    u += g2 * (g0 * d[1][0] + g2 * d[1][1] + g4 * d[1][2]);
    u += g4 * (g0 * d[2][0] + g2 * d[2][1] + g4 * d[2][2]);
    up[0] = u;
    // (even y, odd x): columns b, b+1 under g1, g3
    u = 0.f;
    u += g0 * (g1 * d[0][1] + g3 * d[0][2]);
    u += g2 * (g1 * d[1][1] + g3 * d[1][2]);
    u += g4 * (g1 * d[2][1] + g3 * d[2][2]);
    up[1] = u;
    // (odd y, even x): rows a, a+1 under g1, g3
    u = 0.f;
    u += g1 * (g0 * d[1][0] + g2 * d[1][1] + g4 * d[1][2]);
    u += g3 * (g0 * d[2][0] + g2 * d[2][1] + g4 * d[2][2]);
    up[2] = u;
    u = 0.f;
    u += g1 * (g1 * d[1][1] + g3 * d[1][2]);
    u += g3 * (g1 * d[2][1] + g3 * d[2][2]);
    up[3] = u;
}

// ---- U^T(q) at cell (a, b) of the (h/2 x w/2) plane, U = gauss5 * zero_stuff (reflect); q is (h x w) ------------------------------------------------
MG_STENCIL_FN float upT_general(const float* __restrict__ qp, int a, int b, int h, int w) {
    const float g[5] = MG_G1;
    float acc = 0.f;
    // stuffed index 2a and its reflect pre-images: -2a (a == 1), 2(h-1) - 2a (== h when a == h/2 - 1)
#pragma unroll
    for (int sa = 0; sa < 3; ++sa) {
        int ma = sa == 0 ? 2 * a : (sa == 1 ? -2 * a : 2 * (h - 1) - 2 * a);
        if (sa == 1 && a != 1) continue;
        if (sa == 2 && ma != h) continue;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int y = ma - i + 2;
            if (y < 0 || y >= h) continue;
            // the row's 15 column candidates as one batch of loads (clamped columns), added under the tap-by-tap conditions in the same order
            float v[3][5];
            bool ok[3][5];
#pragma unroll
            for (int sb = 0; sb < 3; ++sb) {
                const int mb = sb == 0 ? 2 * b : (sb == 1 ? -2 * b : 2 * (w - 1) - 2 * b);
                const bool on = !(sb == 1 && b != 1) && !(sb == 2 && mb != w);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int xq = mb - j + 2;
                    ok[sb][j] = on && xq >= 0 && xq < w;
                    v[sb][j] = qp[y * w + (xq < 0 ? 0 : (xq >= w ? w - 1 : xq))];      // unconditional, from a clamped column: used only where ok
                }
            }
            float rowacc = 0.f;
#pragma unroll
            for (int sb = 0; sb < 3; ++sb)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if (ok[sb][j]) rowacc += g[j] * v[sb][j];
                }
            acc += g[i] * rowacc;
        }
    }
    return acc;
}
// inner ring (2, 1, 2, 1): rows 2a+2 ... 2a-2 (i = 0..4), columns 2b+2 ... 2b-2 (j = 0..4)
MG_STENCIL_FN float upT_inner(const float* __restrict__ qp, int a, int b, int w) {
    const float g[5] = MG_G1;
    float v[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float* __restrict__ row = qp + (long)(2 * a + 2 - i) * w + (2 * b - 2);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[i][k] = row[k];
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float rowacc = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) rowacc += g[j] * v[i][4 - j];
        acc += g[i] * rowacc;
    }
    return acc;
}

// ---- D^T(r) at pixel (Y, X) of the (h x w) plane, D = decimate2(gauss5 * .) (reflect); r is (hd x wd) -----------------------------------------------
MG_STENCIL_FN float downT_general(const float* __restrict__ rp, int Y, int X, int h, int w, int hd, int wd) {
    const float g[5] = MG_G1;
    float acc = 0.f;
#pragma unroll
    for (int sa = 0; sa < 3; ++sa) {
        int my = sa == 0 ? Y : (sa == 1 ? -Y : 2 * (h - 1) - Y);
        if (sa == 1 && !(Y == 1 || Y == 2)) continue;
        if (sa == 2 && !(Y == h - 2 || Y == h - 3)) continue;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int ty = my - i + 2;
            if (ty < 0 || (ty & 1)) continue;
            int y = ty >> 1;
            if (y >= hd) continue;
            float v[3][5];
            bool ok[3][5];
#pragma unroll
            for (int sb = 0; sb < 3; ++sb) {
                const int mx = sb == 0 ? X : (sb == 1 ? -X : 2 * (w - 1) - X);
                const bool on = !(sb == 1 && !(X == 1 || X == 2)) && !(sb == 2 && !(X == w - 2 || X == w - 3));
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int tx = mx - j + 2;
                    const int xr = tx >> 1;
                    ok[sb][j] = on && tx >= 0 && !(tx & 1) && xr < wd;
                    v[sb][j] = rp[y * wd + (xr < 0 ? 0 : (xr >= wd ? wd - 1 : xr))];   // unconditional, from a clamped column: used only where ok
                }
            }
            float rowacc = 0.f;
#pragma unroll
            for (int sb = 0; sb < 3; ++sb)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if (ok[sb][j]) rowacc += g[j] * v[sb][j];
                }
            acc += g[i] * rowacc;
        }
    }
    return acc;
}
// the quad (2a + dy, 2b + dx), inner ring (2, 2, 2, 2) of the quad grid: acc[2 * dy + dx]. Even Y: taps i = 0, 2, 4 reach rows a+1, a, a-1; odd Y:
// i = 1, 3 reach rows a+1, a; columns the same way.
MG_STENCIL_FN void downT_inner(const float* __restrict__ rp, int a, int b, int wd, float* acc) {
    const float* __restrict__ r0 = rp + (a - 1) * wd + (b - 1);
    float v[3][3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) v[rr][cc] = r0[rr * wd + cc];
    const float g0 = 1.f / 16.f, g1 = 4.f / 16.f, g2 = 6.f / 16.f, g3 = 4.f / 16.f, g4 = 1.f / 16.f;
    float ra, rb, rc;
    ra = 0.f; ra += g0 * v[2][2]; ra += g2 * v[2][1]; ra += g4 * v[2][0];
    rb = 0.f; rb += g0 * v[1][2]; rb += g2 * v[1][1]; rb += g4 * v[1][0];
    rc = 0.f; rc += g0 * v[0][2]; rc += g2 * v[0][1]; rc += g4 * v[0][0];
    { float s = 0.f; s += g0 * ra; s += g2 * rb; s += g4 * rc; acc[0] = s; }      // (even, even)
    { float s = 0.f; s += g1 * ra; s += g3 * rb; acc[2] = s; }                    // (odd, even): the same row sums under taps 1, 3
    ra = 0.f; ra += g1 * v[2][2]; ra += g3 * v[2][1];                             // odd X: columns b+1, b under g1, g3
    rb = 0.f; rb += g1 * v[1][2]; rb += g3 * v[1][1];
    rc = 0.f; rc += g1 * v[0][2]; rc += g3 * v[0][1];
    { float s = 0.f; s += g0 * ra; s += g2 * rb; s += g4 * rc; acc[1] = s; }      // (even, odd)
    { float s = 0.f; s += g1 * ra; s += g3 * rb; acc[3] = s; }                    // (odd, odd)
}

// ---- adjoint of the replicate-padded Sobel pair at pixel (Y, X): sum over the neighbours (y, x) whose 3 x 3 window reaches (Y, X) -------------------
MG_STENCIL_FN float sobel_adj_general(const float* __restrict__ Ap, const float* __restrict__ Bp, int Y, int X, int H, int W) {
    const float kx[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        int y = Y + dy;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int x = X + dx;
            if (x < 0 || x >= W) continue;
            float av = Ap[y * W + x], bv = Bp[y * W + x];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (st_clampi(y + i - 1, H) != Y) continue;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (st_clampi(x + j - 1, W) != X) continue;
                    acc += kx[i][j] * av + kx[j][i] * bv;
                }
            }
        }
    }
    return acc;
}
// inner ring (1, 1, 1, 1): every neighbour (dy, dx) contributes exactly the term (i, j) = (1 - dy, 1 - dx)
MG_STENCIL_FN float sobel_adj_inner(const float* __restrict__ Ap, const float* __restrict__ Bp, int Y, int X, int W) {
    const float kx[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
    float a[3][3], b[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            a[dy][dx] = Ap[(Y + dy - 1) * W + (X + dx - 1)];
            b[dy][dx] = Bp[(Y + dy - 1) * W + (X + dx - 1)];
        }
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) acc += kx[2 - dy][2 - dx] * a[dy][dx] + kx[2 - dx][2 - dy] * b[dy][dx];
    return acc;
}
