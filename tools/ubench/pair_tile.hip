// Micro-benchmark of the OUTER-PRODUCT register tile for the RotatE pair sum ("next design"):
//   S[i][p] = sum_k | q[i][k] - x[p][k] |        i < B rows, p < P positions, k < d complex dims
// A lane owns a TM x TN block of (row, position) outputs and walks the dims; operands come from LDS in k-pair form
// (re_k, re_k+1, im_k, im_k+1: the packed-op layout of pair_term_cmod2); no cross-lane reduction anywhere.
//   workgroup = 4 waves as 2 x 2 wave tiles; a wave = 8 x 8 lanes; lane (lr, lp) owns rows {lr + 8 a} and positions {lp + 8 b}
//   (interleaved ownership: the 8 lanes of an instruction read 128 contiguous bytes -> no bank conflict)
//   K split over gridDim.z, partial sums stored per split.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pair_tile pair_tile.hip     Run: ./pair_tile [B P d ksplit]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#ifndef TM
#define TM 4
#endif
#ifndef TN
#define TN 4
#endif
#ifndef KUNROLL
#define KUNROLL 1
#endif
constexpr int kUnroll = KUNROLL;          // k pairs whose LDS reads are issued together
constexpr int KC = 16;                    // dims per LDS chunk (8 k-pairs)
constexpr int WROWS = 8 * TM, WPOS = 8 * TN;  // wave tile
constexpr int GROWS = 2 * WROWS, GPOS = 2 * WPOS;  // workgroup tile (2 x 2 waves)
constexpr int PITCH_Q = GROWS * 4 + 4, PITCH_X = GPOS * 4 + 4;  // floats per k-pair row of the LDS image (+16 B pad)

__device__ __forceinline__ f2 pair2(f2 qr, f2 qi, f2 xr, f2 xi) {
    const f2 a = qr - xr, b = qi - xi;
    const f2 n2 = a * a + b * b;
    return f2{__builtin_amdgcn_sqrtf(n2.x), __builtin_amdgcn_sqrtf(n2.y)};
}

struct Args {
    const float *Q;      // [B][2 d]  re | im
    const float *X;      // [N][2 d]
    const int *pool;     // [P] row of X per position
    float *part;         // [ksplit][B][P]
    int B, P, d;
};

__global__ __launch_bounds__(256) void pair_tile_kernel(Args A) {
    __shared__ __attribute__((aligned(16))) float sq[2][KC / 2][PITCH_Q];
    __shared__ __attribute__((aligned(16))) float sx[2][KC / 2][PITCH_X];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 7, lp = lane >> 3;
    const int wr = (wave >> 1) * WROWS, wp = (wave & 1) * WPOS;
    const int i0 = blockIdx.x * GROWS, p0 = blockIdx.y * GPOS;
    const int ks = gridDim.z, kper = ((A.d + ks - 1) / ks + KC - 1) / KC * KC;
    const int k_lo = blockIdx.z * kper, k_hi = min(A.d, k_lo + kper);

    // staging: thread -> (row of the tile, k quad); GROWS * 4 and GPOS * 4 float4 pairs per chunk
    constexpr int QE = GROWS * 4 / 256, XE = GPOS * 4 / 256;  // (row, k quad) items per thread: 1 for 64-row tiles
    float4 rq_re[QE], rq_im[QE], rx_re[XE], rx_im[XE];
    auto gload = [&](int k0) {
#pragma unroll
        for (int e = 0; e < QE; ++e) {
            const int it = tid + e * 256, row = it >> 2, kq = it & 3;
            const int k = min(k0 + 4 * kq, A.d - 4);
            const float *src = A.Q + (size_t)min(i0 + row, A.B - 1) * 2 * A.d + k;
            rq_re[e] = *reinterpret_cast<const float4 *>(src);
            rq_im[e] = *reinterpret_cast<const float4 *>(src + A.d);
        }
#pragma unroll
        for (int e = 0; e < XE; ++e) {
            const int it = tid + e * 256, row = it >> 2, kq = it & 3;
            const int k = min(k0 + 4 * kq, A.d - 4);
            const float *src = A.X + (size_t)A.pool[min(p0 + row, A.P - 1)] * 2 * A.d + k;
            rx_re[e] = *reinterpret_cast<const float4 *>(src);
            rx_im[e] = *reinterpret_cast<const float4 *>(src + A.d);
        }
    };
    auto lstore = [&](int buf, int k0) {
#pragma unroll
        for (int e = 0; e < QE; ++e) {
            const int it = tid + e * 256, row = it >> 2, kq = it & 3;
            const bool ok = k0 + 4 * kq < k_hi;  // (dims past the split's range contribute |0 - 0| = 0)
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 a = ok ? rq_re[e] : z, b = ok ? rq_im[e] : z;
            *reinterpret_cast<float4 *>(&sq[buf][2 * kq][row * 4]) = make_float4(a.x, a.y, b.x, b.y);
            *reinterpret_cast<float4 *>(&sq[buf][2 * kq + 1][row * 4]) = make_float4(a.z, a.w, b.z, b.w);
        }
#pragma unroll
        for (int e = 0; e < XE; ++e) {
            const int it = tid + e * 256, row = it >> 2, kq = it & 3;
            const bool ok = k0 + 4 * kq < k_hi;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 a = ok ? rx_re[e] : z, b = ok ? rx_im[e] : z;
            *reinterpret_cast<float4 *>(&sx[buf][2 * kq][row * 4]) = make_float4(a.x, a.y, b.x, b.y);
            *reinterpret_cast<float4 *>(&sx[buf][2 * kq + 1][row * 4]) = make_float4(a.z, a.w, b.z, b.w);
        }
    };

    f2 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f2{0.f, 0.f};

    gload(k_lo);
    lstore(0, k_lo);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
        const bool more = k0 + KC < k_hi;
        if (more) gload(k0 + KC);
#pragma unroll kUnroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            float4 q[TM], x[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) q[a] = *reinterpret_cast<const float4 *>(&sq[buf][kp][(wr + lr + 8 * a) * 4]);
#pragma unroll
            for (int b = 0; b < TN; ++b) x[b] = *reinterpret_cast<const float4 *>(&sx[buf][kp][(wp + lp + 8 * b) * 4]);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] += pair2(f2{q[a].x, q[a].y}, f2{q[a].z, q[a].w}, f2{x[b].x, x[b].y}, f2{x[b].z, x[b].w});
        }
        if (more) lstore(buf ^ 1, k0 + KC);
        __syncthreads();
        buf ^= 1;
    }
    float *out = A.part + (size_t)blockIdx.z * A.B * A.P;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int i = i0 + wr + lr + 8 * a;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int p = p0 + wp + lp + 8 * b;
            if (i < A.B && p < A.P) out[(size_t)i * A.P + p] = acc[a][b].x + acc[a][b].y;
        }
    }
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024, P = argc > 2 ? atoi(argv[2]) : 256, d = argc > 3 ? atoi(argv[3]) : 1000;
    const int ks = argc > 4 ? atoi(argv[4]) : 8, N = 14541;
    std::vector<float> hq((size_t)B * 2 * d), hx((size_t)N * 2 * d);
    std::vector<int> hp(P);
    srand(1);
    for (auto &v : hq) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    for (auto &v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    for (auto &v : hp) v = rand() % N;
    float *dq, *dx, *dpart;
    int *dp;
    hipMalloc(&dq, hq.size() * 4); hipMalloc(&dx, hx.size() * 4); hipMalloc(&dp, P * 4);
    hipMalloc(&dpart, (size_t)ks * B * P * 4);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, hp.data(), P * 4, hipMemcpyHostToDevice);
    Args A{dq, dx, dp, dpart, B, P, d};
    dim3 grid((B + GROWS - 1) / GROWS, (P + GPOS - 1) / GPOS, ks);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(pair_tile_kernel, grid, dim3(256), 0, 0, A);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(pair_tile_kernel, grid, dim3(256), 0, 0, A);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> part((size_t)ks * B * P);
    hipMemcpy(part.data(), dpart, part.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 64; ++t) {  // spot check against a double-precision sum
        const int i = (t * 131) % B, p = (t * 37) % P;
        double ref = 0;
        for (int k = 0; k < d; ++k) {
            const double a = hq[(size_t)i * 2 * d + k] - hx[(size_t)hp[p] * 2 * d + k];
            const double b = hq[(size_t)i * 2 * d + d + k] - hx[(size_t)hp[p] * 2 * d + d + k];
            ref += sqrt(a * a + b * b);
        }
        double got = 0;
        for (int z = 0; z < ks; ++z) got += part[((size_t)z * B + i) * P + p];
        worst = fmax(worst, fabs(got - ref));
    }
    const double us = ms * 1e3 / reps, terms = (double)B * P * d;
    printf("TM=%d TN=%d B=%d P=%d d=%d ksplit=%d grid=%dx%dx%d: %.1f us per launch, %.2f cycles per wave-term @2.4GHz (1024 SIMDs), "
           "%.1f TFLOP/s at 6 flop per term, max abs err %.2e\n", TM, TN, B, P, d, ks, grid.x, grid.y, grid.z, us,
           us * 1e-6 * 2.4e9 * 1024 / (terms / 64), terms * 6 / (us * 1e-6) / 1e12, worst);
    return worst < 1e-3 ? 0 : 1;
}
