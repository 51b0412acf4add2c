// Micro-benchmark of the outer-product register tile for the BACKWARD of the RotatE pair sum:
//   w = G[i][p] / |q[i][k] - x[p][k]|,   dQ[i][k] -= w (q - x),   dX[p][k] += w (q - x)      (complex k, both halves)
// workgroup = (tile of 64 rows, chunk of 16 dims); it walks ALL position tiles of 64, so dQ of its rows / dims is complete in
// registers and written once; dX of a position tile is partial over the workgroup's 64 rows: [row tile][P][2 d] partials.
// A lane owns 4 x 4 (row, position) pairs (rows lr + 8a, positions lp + 8c of its wave's 32 x 32 tile); per pair of dims it
// evaluates 16 pair terms, then
//   dq partials (4 rows x 4 floats) are summed over the 8 lanes that share lr: two transposed permlane-swap levels (the data
//     halves each time) + one DPP add: afterwards lane (lp) holds row (lp >> 1) & 3 ... see reduce_dq;
//   dx partials (4 positions x 4 floats) are summed over the 8 lanes that share lp: three DPP adds each (all lanes get all).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pair_tile_bwd pair_tile_bwd.hip     Run: ./pair_tile_bwd [B P d]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int KC = 16, ROWS = 64, POS = 64, PITCH = ROWS * 4 + 4;

struct Args {
    const float *Q, *X, *G;  // Q [B][2d], X [N][2d], G [B][P]
    const int *pool;
    float *dQ;               // [B][2d]
    float *dXp;              // [row tiles][P][2d]
    int B, P, d;
};


// sum over lanes that differ in lane bits 0..2 (the 8 lanes sharing lp); every lane gets the total
__device__ __forceinline__ float sum_lr(float t) {
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x141, 0xf, 0xf, false));  // row_half_mirror
    return t;
}

// v[16] per-lane partials -> r[4]: sums over the 8 lanes sharing lr (lane bits 3, 4, 5).  Transposed: after the 32-swap lanes
// 0-31 keep v[0..7] (+ partner), lanes 32-63 keep v[8..15]; after the 16-swap even 16-lane rows keep the first half of
// those, odd rows the second; the xor-8 level is a plain DPP add.  Value held by r[j]: index 8 * (lane >> 5) + 4 * ((lane >> 4) & 1) + j.
__device__ __forceinline__ void reduce_dq(const float (&v)[16], float (&r)[4]) {
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 8]), false, false);
        w[j] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[j]), __float_as_uint(w[j + 4]), false, false);
        float t = __uint_as_float(s[0]) + __uint_as_float(s[1]);
        t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x128, 0xf, 0xf, false));  // row_ror:8
        r[j] = t;
    }
}

__global__ __launch_bounds__(256) void pair_tile_bwd_kernel(Args A) {
    __shared__ __attribute__((aligned(16))) float sq[KC / 2][PITCH];
    __shared__ __attribute__((aligned(16))) float sx[2][KC / 2][PITCH];
    __shared__ __attribute__((aligned(16))) float sdx[4][32][KC / 2][4];  // per wave: dx of its 32 positions x 8 dim pairs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 7, lp = lane >> 3;
    const int wr = (wave >> 1) * 32, wp = (wave & 1) * 32;
    const int i0 = blockIdx.x * ROWS, k0 = blockIdx.y * KC, d = A.d;
    const int srow = tid >> 2, skq = tid & 3;
    const int kk = min(k0 + 4 * skq, d - 4);
    const bool kok = k0 + 4 * skq < d;
    const float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
    {   // q chunk of the row tile: staged once
        const float *src = A.Q + (size_t)min(i0 + srow, A.B - 1) * 2 * d + kk;
        const float4 a = kok ? *reinterpret_cast<const float4 *>(src) : zz, b = kok ? *reinterpret_cast<const float4 *>(src + d) : zz;
        *reinterpret_cast<float4 *>(&sq[2 * skq][srow * 4]) = make_float4(a.x, a.y, b.x, b.y);
        *reinterpret_cast<float4 *>(&sq[2 * skq + 1][srow * 4]) = make_float4(a.z, a.w, b.z, b.w);
    }
    float4 rx_re, rx_im;
    auto gload = [&](int p0) {
        const float *src = A.X + (size_t)A.pool[min(p0 + srow, A.P - 1)] * 2 * d + kk;
        rx_re = *reinterpret_cast<const float4 *>(src);
        rx_im = *reinterpret_cast<const float4 *>(src + d);
    };
    auto lstore = [&](int buf) {
        const float4 a = kok ? rx_re : zz, b = kok ? rx_im : zz;
        *reinterpret_cast<float4 *>(&sx[buf][2 * skq][srow * 4]) = make_float4(a.x, a.y, b.x, b.y);
        *reinterpret_cast<float4 *>(&sx[buf][2 * skq + 1][srow * 4]) = make_float4(a.z, a.w, b.z, b.w);
    };
    // dq of the workgroup's rows / dims accumulates in LDS: after reduce_dq a lane holds the 4 floats (re_k, re_k+1, im_k,
    // im_k+1) of ONE row -- a = 2 * (lane >> 5) + ((lane >> 4) & 1) -- so the update is one 16-byte read-modify-write per lane
    // and dim pair, owned by that lane alone (lanes that differ in bit 3 hold duplicates: the lower one writes)
    __shared__ __attribute__((aligned(16))) float sdq[4][32][KC / 2][4];
    for (int e = tid; e < 4 * 32 * (KC / 2); e += 256) *reinterpret_cast<float4 *>(&sdq[0][0][0][0] + 4 * e) = zz;
    const int dq_row = lr + 8 * (2 * (lane >> 5) + ((lane >> 4) & 1));  // row of the wave's 32 this lane accumulates
    const int n_pt = (A.P + POS - 1) / POS;
    gload(0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int pt = 0; pt < n_pt; ++pt) {
        const int p0 = pt * POS;
        if (pt + 1 < n_pt) gload(p0 + POS);
        float g[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = i0 + wr + lr + 8 * a, p = p0 + wp + lp + 8 * c;
                g[a][c] = (i < A.B && p < A.P) ? A.G[(size_t)i * A.P + p] : 0.f;
            }
#pragma unroll 1
        for (int kp = 0; kp < KC / 2; ++kp) {
            float4 q[4], x[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) q[a] = *reinterpret_cast<const float4 *>(&sq[kp][(wr + lr + 8 * a) * 4]);
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = *reinterpret_cast<const float4 *>(&sx[buf][kp][(wp + lp + 8 * c) * 4]);
            // positions outer, rows inner: a position's dx is complete after its 4 rows and goes through its reduction at
            // once (2 live f2 instead of 8); dq of the 4 rows accumulates across the positions
            f2 dqr[4], dqi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { dqr[j] = f2{0.f, 0.f}; dqi[j] = f2{0.f, 0.f}; }
            const f2 eps = f2{1e-30f, 1e-30f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f2 dxr = f2{0.f, 0.f}, dxi = f2{0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f2 da = f2{q[a].x, q[a].y} - f2{x[c].x, x[c].y}, db = f2{q[a].z, q[a].w} - f2{x[c].z, x[c].w};
                    const f2 n2 = __builtin_elementwise_fma(db, db, __builtin_elementwise_fma(da, da, eps));
                    const f2 w = f2{__builtin_amdgcn_rsqf(n2.x), __builtin_amdgcn_rsqf(n2.y)} * g[a][c];
                    dqr[a] = __builtin_elementwise_fma(-w, da, dqr[a]);
                    dqi[a] = __builtin_elementwise_fma(-w, db, dqi[a]);
                    dxr = __builtin_elementwise_fma(w, da, dxr);
                    dxi = __builtin_elementwise_fma(w, db, dxi);
                }
#ifdef NO_DXRED
                const float4 u = make_float4(dxr.x, dxr.y, dxi.x, dxi.y);
#else
                const float4 u = make_float4(sum_lr(dxr.x), sum_lr(dxr.y), sum_lr(dxi.x), sum_lr(dxi.y));
#endif
#ifdef NO_DXLDS
                if (u.x == 12345.f) sdx[wave][0][0][0] = u.y + u.z + u.w;
#else
                if (lr == kp) *reinterpret_cast<float4 *>(&sdx[wave][lp + 8 * c][kp][0]) = u;
#endif
            }
            float v[16], r[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { v[4 * a] = dqr[a].x; v[4 * a + 1] = dqr[a].y; v[4 * a + 2] = dqi[a].x; v[4 * a + 3] = dqi[a].y; }
#ifdef NO_DQRED
            r[0] = v[0] + v[4] + v[8] + v[12]; r[1] = v[1] + v[5] + v[9] + v[13]; r[2] = v[2] + v[6] + v[10] + v[14]; r[3] = v[3] + v[7] + v[11] + v[15];
#else
            reduce_dq(v, r);
#endif
            if ((lane & 8) == 0) {
                float4 t = *reinterpret_cast<const float4 *>(&sdq[wave][dq_row][kp][0]);
                t.x += r[0]; t.y += r[1]; t.z += r[2]; t.w += r[3];
                *reinterpret_cast<float4 *>(&sdq[wave][dq_row][kp][0]) = t;
            }
        }
        if (pt + 1 < n_pt) lstore(buf ^ 1);
        __syncthreads();
        // the two waves of a column (rows 0-31 / 32-63) add up; 64 positions x 16 dims x (re, im) go out as 16-byte stores
        for (int e = tid; e < 2 * 32 * 4; e += 256) {
            const int wc = e >> 7, pl = (e >> 2) & 31, kq = e & 3;
            const float4 a0 = *reinterpret_cast<const float4 *>(&sdx[wc][pl][2 * kq][0]), a1 = *reinterpret_cast<const float4 *>(&sdx[wc][pl][2 * kq + 1][0]);
            const float4 b0 = *reinterpret_cast<const float4 *>(&sdx[2 + wc][pl][2 * kq][0]), b1 = *reinterpret_cast<const float4 *>(&sdx[2 + wc][pl][2 * kq + 1][0]);
            const int p = p0 + wc * 32 + pl, k = k0 + 4 * kq;
            if (p < A.P && k < d) {
                float *dst = A.dXp + ((size_t)blockIdx.x * A.P + p) * 2 * d + k;
                *reinterpret_cast<float4 *>(dst) = make_float4(a0.x + b0.x, a0.y + b0.y, a1.x + b1.x, a1.y + b1.y);
                *reinterpret_cast<float4 *>(dst + d) = make_float4(a0.z + b0.z, a0.w + b0.w, a1.z + b1.z, a1.w + b1.w);
            }
        }
        __syncthreads();
        buf ^= 1;
    }
    // dq: the two waves of a row block (positions 0-31 / 32-63 of every tile) add up; 16-byte stores of 4 dims per half
    __syncthreads();
    for (int e = tid; e < 2 * 32 * 4; e += 256) {
        const int wrow = e >> 7, rl = (e >> 2) & 31, kq = e & 3;
        const float4 a0 = *reinterpret_cast<const float4 *>(&sdq[2 * wrow][rl][2 * kq][0]), a1 = *reinterpret_cast<const float4 *>(&sdq[2 * wrow][rl][2 * kq + 1][0]);
        const float4 b0 = *reinterpret_cast<const float4 *>(&sdq[2 * wrow + 1][rl][2 * kq][0]), b1 = *reinterpret_cast<const float4 *>(&sdq[2 * wrow + 1][rl][2 * kq + 1][0]);
        const int i = i0 + 32 * wrow + rl, k = k0 + 4 * kq;
        if (i < A.B && k < d) {
            float *dst = A.dQ + (size_t)i * 2 * d + k;
            *reinterpret_cast<float4 *>(dst) = make_float4(a0.x + b0.x, a0.y + b0.y, a1.x + b1.x, a1.y + b1.y);
            *reinterpret_cast<float4 *>(dst + d) = make_float4(a0.z + b0.z, a0.w + b0.w, a1.z + b1.z, a1.w + b1.w);
        }
    }
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024, P = argc > 2 ? atoi(argv[2]) : 256, d = argc > 3 ? atoi(argv[3]) : 1000, N = 14541;
    std::vector<float> hq((size_t)B * 2 * d), hx((size_t)N * 2 * d), hg((size_t)B * P);
    std::vector<int> hp(P);
    srand(1);
    for (auto &v : hq) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    for (auto &v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    for (auto &v : hg) v = rand() / (float)RAND_MAX * 1e-3f;
    for (auto &v : hp) v = rand() % N;
    const int rt = (B + ROWS - 1) / ROWS;
    float *dq, *dx, *dg, *ddq, *ddx;
    int *dp;
    hipMalloc(&dq, hq.size() * 4); hipMalloc(&dx, hx.size() * 4); hipMalloc(&dg, hg.size() * 4); hipMalloc(&dp, P * 4);
    hipMalloc(&ddq, hq.size() * 4); hipMalloc(&ddx, (size_t)rt * P * 2 * d * 4);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, hp.data(), P * 4, hipMemcpyHostToDevice);
    Args A{dq, dx, dg, dp, ddq, ddx, B, P, d};
    dim3 grid(rt, (d + KC - 1) / KC);
    hipMemset(ddq, 0, hq.size() * 4);
    hipLaunchKernelGGL(pair_tile_bwd_kernel, grid, dim3(256), 0, 0, A);
    std::vector<float> rdq(hq.size()), rdx((size_t)rt * P * 2 * d);
    hipMemcpy(rdq.data(), ddq, rdq.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rdx.data(), ddx, rdx.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int t = 0; t < 24; ++t) {  // spot checks: dQ[i][k] and dX[p][k] (sum of the row-tile partials)
        const int i = (t * 131) % B, p = (t * 37) % P, k = (t * 71) % d;
        double rq_re = 0, rq_im = 0, rx_re = 0, rx_im = 0;
        for (int pp = 0; pp < P; ++pp) {
            const double a = hq[(size_t)i * 2 * d + k] - hx[(size_t)hp[pp] * 2 * d + k], b = hq[(size_t)i * 2 * d + d + k] - hx[(size_t)hp[pp] * 2 * d + d + k];
            const double w = hg[(size_t)i * P + pp] / sqrt(a * a + b * b + 1e-30);
            rq_re -= w * a; rq_im -= w * b;
        }
        for (int ii = 0; ii < B; ++ii) {
            const double a = hq[(size_t)ii * 2 * d + k] - hx[(size_t)hp[p] * 2 * d + k], b = hq[(size_t)ii * 2 * d + d + k] - hx[(size_t)hp[p] * 2 * d + d + k];
            const double w = hg[(size_t)ii * P + p] / sqrt(a * a + b * b + 1e-30);
            rx_re += w * a; rx_im += w * b;
        }
        double gx_re = 0, gx_im = 0;
        for (int r = 0; r < rt; ++r) { gx_re += rdx[((size_t)r * P + p) * 2 * d + k]; gx_im += rdx[((size_t)r * P + p) * 2 * d + d + k]; }
        worst = fmax(worst, fmax(fabs(rdq[(size_t)i * 2 * d + k] - rq_re), fabs(rdq[(size_t)i * 2 * d + d + k] - rq_im)));
        worst = fmax(worst, fmax(fabs(gx_re - rx_re), fabs(gx_im - rx_im)));
        scale = fmax(scale, fmax(fabs(rq_re), fabs(rx_re)));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(pair_tile_bwd_kernel, grid, dim3(256), 0, 0, A);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("bwd tile B=%d P=%d d=%d grid=%dx%d: %.1f us per launch, max abs err %.2e (scale %.2e)\n", B, P, d, grid.x, grid.y,
           ms * 1e3 / reps, worst, scale);
    return worst < 1e-3 * (scale + 1e-9) + 1e-7 ? 0 : 1;
}
