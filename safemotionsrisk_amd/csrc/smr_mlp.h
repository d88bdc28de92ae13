// smr_mlp.h - batched forward pass of a small fully connected network on the MFMA units (gfx950), for the backup-policy /
// risk-network part of the step path (safe_motions_base.py:1498-1659, actions.py:303-340): the risk network
// (state + action -> risk, train_risk_network.py:273-282) and the backup policy (model/keras_fcnet_last_layer_activation.py:
// observations -> [tanh means | log stds], deterministic action = the means, safe_motions_base.py:631) are dense layers of width
// <= 256 over a batch of 65 536 rows - the one GEMM-shaped piece of this path.
//
// One wavefront carries 16 rows through ALL layers: activations live in two LDS buffers of the wave (16 x (K + 4) floats: the + 4
// puts the 16 rows x 4 k of an A operand in 64 different banks), weights stream from L2 as B operands,
// v_mfma_f32_16x16x4_f32 accumulates (exact f32: A[l&15][k=l>>4], B[k=l>>4][l&15], D col = l&15, row = 4 (l>>4) + reg), four
// independent accumulators per wave cover its 40-cycle dependent latency.  Weights are zero-padded on the host to K multiples of 4
// and N multiples of 64 (16 for the last layer), so the loops need no bounds tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smr {

constexpr int kMlpMaxLayers = 4;
constexpr int kMlpMaxWidth = 256;                      // hidden / input widths (padded)
constexpr int kMlpRows = 16;                           // rows per wavefront
constexpr int kMlpPitch = kMlpMaxWidth + 4;

enum { kActLinear = 0, kActTanh = 1, kActSwish = 2, kActRelu = 3, kActSigmoid = 4, kActSelu = 5 };

struct DMlp {
    int n_layers, in_size, out_size;
    int K[kMlpMaxLayers], N[kMlpMaxLayers], act[kMlpMaxLayers];   // padded sizes
    const float *W[kMlpMaxLayers];                     // [K][N] row major (a Keras Dense kernel), zero padded
    const float *b[kMlpMaxLayers];                     // [N]
};

typedef float mlp_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mlp_activation(float x, int act) {
    switch (act) {
        case kActTanh: return tanhf(x);
        case kActSwish: return x / (1.0f + expf(-x));                       // tf.nn.swish = x sigmoid(x)
        case kActRelu: return fmaxf(x, 0.f);
        case kActSigmoid: return 1.0f / (1.0f + expf(-x));
        case kActSelu: return x > 0.f ? 1.05070098735548f * x : 1.05070098735548f * 1.67326324235438f * (expf(x) - 1.0f);
        default: return x;
    }
}

// x: [n_rows][ldx] (the first in_size columns are read), y: [n_rows][ldy] (out_size columns are written)
// one wavefront per block (33 KB of LDS: four blocks per CU, one per SIMD)
__global__ __launch_bounds__(64) void mlp_forward_kernel(DMlp m, const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy, int n_rows) {
    __shared__ float lds[2][kMlpRows * kMlpPitch];
    const int lane = (int)threadIdx.x;
    const int row0 = (int)blockIdx.x * kMlpRows;
    if (row0 >= n_rows) return;
    float *in = lds[0], *out = lds[1];
    const int r = lane & 15, kq = lane >> 4;
    // the rows' inputs, zero padded to K[0]
    for (int i = lane; i < kMlpRows * m.K[0]; i += 64) {
        const int rr = i / m.K[0], c = i - rr * m.K[0];
        in[rr * kMlpPitch + c] = (row0 + rr < n_rows && c < m.in_size) ? x[(size_t)(row0 + rr) * ldx + c] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int l = 0; l < m.n_layers; ++l) {
        const int K = m.K[l], N = m.N[l];
        const float *__restrict__ W = m.W[l];
        const bool last = l == m.n_layers - 1;
        for (int n0 = 0; n0 < N; n0 += 64) {
            const int tiles = (N - n0) >= 64 ? 4 : (N - n0) / 16;           // N is a multiple of 16
            mlp_f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 8
            for (int k0 = 0; k0 < K; k0 += 4) {         // unrolled: the B operands of eight k steps are in flight together
                const float a = in[r * kMlpPitch + k0 + kq];
                const float *wrow = W + (size_t)(k0 + kq) * N + n0 + r;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < tiles) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wrow[16 * t], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= tiles) continue;
                const int col = n0 + 16 * t + r;
                const float bias = m.b[l][col];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rr = 4 * kq + q;
                    const float val = mlp_activation(acc[t][q] + bias, m.act[l]);
                    if (last) { if (col < m.out_size && row0 + rr < n_rows) y[(size_t)(row0 + rr) * ldy + col] = val; }
                    else out[rr * kMlpPitch + col] = val;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float *tmp = in; in = out; out = tmp;
    }
}

}  // namespace smr
