// smr_mlp.h - batched forward pass of a small fully connected network on the MFMA units (gfx950), for the backup-policy /
// risk-network part of the step path (safe_motions_base.py:1498-1659, actions.py:303-340): the risk network
// (state + action -> risk, train_risk_network.py:273-282) and the backup policy (model/keras_fcnet_last_layer_activation.py:
// observations -> [tanh means | log stds], deterministic action = the means, safe_motions_base.py:631) are dense layers of width
// <= 256 over a batch of 65 536 rows - the one GEMM-shaped piece of this path.
//
// One wavefront carries 32 rows through ALL layers.  The rows' activations live in ONE LDS buffer of the wave (32 x (256 + 4)
// floats: the + 4 puts the 16 rows x 4 k of an A operand in 64 different banks); a layer's whole output (32 rows x up to 256
// columns = up to 32 tiles of 16 x 16) accumulates in registers over a single pass down K and replaces the input in place.  Per
// k step of 4 the wave reads two A operands from LDS and one B operand per column tile from L2 (weights: <= 256 KB per layer, every
// wave reads all of them) and issues 2 x N/16 independent v_mfma_f32_16x16x4_f32 (exact f32: A[l&15][k=l>>4], B[k=l>>4][l&15],
// D col = l&15, row = 4 (l>>4) + reg; 32 cycles each on a SIMD) - enough to cover the L2 latency of the next k step's B operands,
// which are fetched one step ahead.
// Contract of mlp_forward_kernel for a DMlp: K[l] % 8 == 0 (the k loop fetches two MFMA k steps together), N[l] % 64 == 0 (four
// 16-column tiles per pass), K[l], N[l] <= kMlpMaxWidth, K[l + 1] == N[l], weights / biases zero padded up to those sizes.
// smr_mlp_create pads K[0] to a multiple of 16 and N to multiples of 64; smr_mlp_forward rejects anything else.  The input rows may
// be wider than in_size (row stride ldx >= in_size: the backup policy reads the first columns of the risk network's rows): only
// columns < in_size are loaded, the padding columns up to K[0] read as zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smr {

constexpr int kMlpMaxLayers = 4;
constexpr int kMlpMaxWidth = 256;                      // hidden / input widths (padded)
constexpr int kMlpRows = 32;                           // rows per wavefront (two 16-row A tiles)
constexpr int kMlpPitch = kMlpMaxWidth + 4;

enum { kActLinear = 0, kActTanh = 1, kActSwish = 2, kActRelu = 3, kActSigmoid = 4, kActSelu = 5 };

struct DMlp {
    int n_layers, in_size, out_size;
    int K[kMlpMaxLayers], N[kMlpMaxLayers], act[kMlpMaxLayers];   // padded sizes
    const float *W[kMlpMaxLayers];                     // [K][N] row major (a Keras Dense kernel), zero padded
    const float *b[kMlpMaxLayers];                     // [N]
};

typedef float mlp_f4 __attribute__((ext_vector_type(4)));

// hardware exp2 / reciprocal (1 ulp each): absolute error of tanh and sigmoid ~2e-7, the level at which float tanh implementations
// (numpy, TensorFlow) differ from each other; the libm calls cost ~40 instructions per value, as much as the layer's MFMAs
__device__ __forceinline__ float mlp_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896f); }
__device__ __forceinline__ float mlp_activation(float x, int act) {
    switch (act) {
        case kActTanh: { const float e = mlp_exp(-2.0f * fabsf(x)); return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x); }
        case kActSwish: return x * __builtin_amdgcn_rcpf(1.0f + mlp_exp(-x));      // tf.nn.swish = x sigmoid(x)
        case kActRelu: return fmaxf(x, 0.f);
        case kActSigmoid: return __builtin_amdgcn_rcpf(1.0f + mlp_exp(-x));
        case kActSelu: return x > 0.f ? 1.05070098735548f * x : 1.05070098735548f * 1.67326324235438f * (mlp_exp(x) - 1.0f);
        default: return x;
    }
}

typedef const __attribute__((address_space(1))) float *mlp_gptr;   // global address space: a pointer out of a struct is generic, its
                                                                    // loads are flat loads and count against the LDS counter too

// one layer of one wave: act[32][K] (LDS) -> act[32][N] in place, or -> y for the last layer.  NP = N / 64.
template <int NP>
__device__ __forceinline__ void mlp_layer(float *act_lds, mlp_gptr W, mlp_gptr bias, int K, int act, bool last, int out_size,
                                          float *__restrict__ y, int ldy, int row0, int n_rows) {
    constexpr int NT = 4 * NP, N = 64 * NP;
    const int lane = (int)threadIdx.x, r = lane & 15, kq = lane >> 4;
    mlp_f4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[0][t] = acc[1][t] = mlp_f4{0, 0, 0, 0};
    float b0[NT], b1[NT];
    mlp_gptr wlane = W + (size_t)kq * N + r;
#pragma unroll
    for (int t = 0; t < NT; ++t) b0[t] = wlane[16 * t];
    const float *alane = act_lds + r * kMlpPitch + kq;
    for (int k0 = 0; k0 < K; k0 += 8) {
        mlp_gptr w1 = wlane + (size_t)(k0 + 4) * N;
#pragma unroll
        for (int t = 0; t < NT; ++t) b1[t] = w1[16 * t];
        float a0 = alane[k0], a1 = alane[16 * kMlpPitch + k0];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[t], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0[t], acc[1][t], 0, 0, 0);
        }
        // left alone the scheduler sinks every load down to the MFMA that uses it (load, wait, two MFMAs, load, wait ...): one weight
        // load of the NEXT k step goes between each pair of MFMAs of this one
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one VMEM read
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // two MFMAs
        }
        const int kn = k0 + 8 < K ? k0 + 8 : k0;         // the last step re-reads a row it has: no branch around the loads
        mlp_gptr w0 = wlane + (size_t)kn * N;
#pragma unroll
        for (int t = 0; t < NT; ++t) b0[t] = w0[16 * t];
        a0 = alane[k0 + 4]; a1 = alane[16 * kMlpPitch + k0 + 4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1[t], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[t], acc[1][t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
    }
    // every read of the input is done (same wave, program order): the output goes where the input was, as acc + bias; the activation
    // is a second, rolled-up pass over the buffer (the unrolled tile loop below is 8 NT values long - an activation inlined here would
    // be most of the kernel's code) and the last layer's pass writes y with consecutive lanes on consecutive columns
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = 16 * t + r;
        const float bv = bias[col];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 4; ++q) act_lds[(16 * h + 4 * kq + q) * kMlpPitch + col] = acc[h][t][q] + bv;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (last) {
        for (int i = lane; i < kMlpRows * out_size; i += 64) {
            const int rr = i / out_size, c = i - rr * out_size;
            if (row0 + rr < n_rows) y[(size_t)(row0 + rr) * ldy + c] = mlp_activation(act_lds[rr * kMlpPitch + c], act);
        }
    } else if (act != kActLinear) {
        for (int rr = 0; rr < kMlpRows; ++rr)
            for (int c = lane; c < N; c += 64) act_lds[rr * kMlpPitch + c] = mlp_activation(act_lds[rr * kMlpPitch + c], act);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// x: [n_rows][ldx] (the first in_size columns are read), y: [n_rows][ldy] (out_size columns are written)
// one wavefront per block (33 KB of LDS: four blocks per CU, one per SIMD)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void mlp_forward_kernel(DMlp m, const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy, int n_rows) {
    __shared__ float lds[kMlpRows * kMlpPitch];
    const int lane = (int)threadIdx.x;
    const int row0 = (int)blockIdx.x * kMlpRows;
    if (row0 >= n_rows) return;
    // the rows' inputs, zero padded to K[0]
    for (int i = lane; i < kMlpRows * m.K[0]; i += 64) {
        const int rr = i / m.K[0], c = i - rr * m.K[0];
        lds[rr * kMlpPitch + c] = (row0 + rr < n_rows && c < m.in_size) ? x[(size_t)(row0 + rr) * ldx + c] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int l = 0; l < m.n_layers; ++l) {
        const bool last = l == m.n_layers - 1;
        mlp_gptr W = (mlp_gptr)m.W[l], bias = (mlp_gptr)m.b[l];
        switch (m.N[l] >> 6) {
            case 1: mlp_layer<1>(lds, W, bias, m.K[l], m.act[l], last, m.out_size, y, ldy, row0, n_rows); break;
            case 2: mlp_layer<2>(lds, W, bias, m.K[l], m.act[l], last, m.out_size, y, ldy, row0, n_rows); break;
            case 3: mlp_layer<3>(lds, W, bias, m.K[l], m.act[l], last, m.out_size, y, ldy, row0, n_rows); break;
            default: mlp_layer<4>(lds, W, bias, m.K[l], m.act[l], last, m.out_size, y, ldy, row0, n_rows); break;
        }
    }
}

}  // namespace smr
