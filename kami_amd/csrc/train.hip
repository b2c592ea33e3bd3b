// train.hip — NN::train (kami/nn/nn.cpp:224-377) on the device: SGD steps through the same network
// (nn.cpp:59-91) in TRAINING mode — BatchNorm on batch statistics with the running statistics
// updated (momentum 0.1, unbiased variance), the reference's loss (nn.cpp:93-105)
//     L = -sum(obs_p * log(p + 0.001)) + mean((v - obs_v)^2)      (v is [B,256], obs_v broadcasts: Q10)
// and plain SGD p -= lr * dL/dp.
//
// First correct path: fp32 VALU kernels, one launch per operation, every
// reduction done by one workgroup in a fixed order (deterministic).  At the reference's batch of 8
// the work per step is a few MFLOP and the step is launch-bound (~300 launches); batching the
// training set into MFMA kernels is the next step, not this file's.
//
// Parameters live on the device in the canonical blob order (kh_weight_count in kami_hip.h), with
// libtorch's own tensor shapes: conv weight [Co][Ci][3][3] / [Co][Ci][1][1].  Gradients use the same
// layout (the BatchNorm running-statistics slots stay zero), so SGD is one axpy over the blob.
// Activations are [B][64][C] fp32 channels-last like forward_simple.hip.
#include "kh_internal.h"

#include <vector>

namespace kh {

namespace {

constexpr float BN_EPS = 1e-5f, BN_MOMENTUM = 0.1f;

// ---- conv ----------------------------------------------------------------------------------------
// y[b][p][co] = bias[co] + sum_{tap,ci} x[b][p + off(tap)][ci] * w[co][ci][tap]
__global__ __launch_bounds__(256) void conv_fwd_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ x, float* __restrict__ y,
                                                       int B, int Ci, int Co, int T)
{
    const long total = (long)B * 64 * Co;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Co), pix = (int)((idx / Co) & 63);
        const long b = idx / ((long)Co * 64);
        const int py = pix >> 3, px = pix & 7;
        float acc = 0.0f;
        for (int t = 0; t < T; ++t) {
            const int iy = T == 9 ? py + t / 3 - 1 : py, ix = T == 9 ? px + t % 3 - 1 : px;
            if (iy < 0 || iy > 7 || ix < 0 || ix > 7) continue;
            const float* xi = x + (b * 64 + iy * 8 + ix) * Ci;
            const float* wk = w + (size_t)co * Ci * T + t;
            for (int ci = 0; ci < Ci; ++ci) acc = fmaf(xi[ci], wk[(size_t)ci * T], acc);
        }
        y[idx] = acc + bias[co];
    }
}

// dx[b][p][ci] (+)= sum_{tap,co} dy[b][p - off(tap)][co] * w[co][ci][tap]
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int B, int Ci, int Co, int T, int accumulate)
{
    const long total = (long)B * 64 * Ci;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % Ci), pix = (int)((idx / Ci) & 63);
        const long b = idx / ((long)Ci * 64);
        const int py = pix >> 3, px = pix & 7;
        float acc = 0.0f;
        for (int t = 0; t < T; ++t) {
            // output pixel q read input pixel q + off(t); this input pixel p feeds q = p - off(t)
            const int oy = T == 9 ? py - (t / 3 - 1) : py, ox = T == 9 ? px - (t % 3 - 1) : px;
            if (oy < 0 || oy > 7 || ox < 0 || ox > 7) continue;
            const float* g = dy + (b * 64 + oy * 8 + ox) * Co;
            const float* wk = w + (size_t)ci * T + t;
            for (int co = 0; co < Co; ++co) acc = fmaf(g[co], wk[(size_t)co * Ci * T], acc);
        }
        dx[idx] = accumulate ? dx[idx] + acc : acc;
    }
}

// dw[co][ci][tap] = sum_{b,p} dy[b][p][co] * x[b][p + off(tap)][ci];  db[co] = sum dy[b][p][co]
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         float* __restrict__ dw, float* __restrict__ db,
                                                         int B, int Ci, int Co, int T)
{
    const long total = (long)Co * Ci * T;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T), ci = (int)((idx / T) % Ci), co = (int)(idx / ((long)T * Ci));
        const int dyy = T == 9 ? t / 3 - 1 : 0, dxx = T == 9 ? t % 3 - 1 : 0;
        float acc = 0.0f, bacc = 0.0f;
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < 64; ++p) {
                const float g = dy[((long)b * 64 + p) * Co + co];
                bacc += g;
                const int iy = (p >> 3) + dyy, ix = (p & 7) + dxx;
                if (iy < 0 || iy > 7 || ix < 0 || ix > 7) continue;
                acc = fmaf(g, x[((long)b * 64 + iy * 8 + ix) * Ci + ci], acc);
            }
        dw[idx] = acc;
        if (ci == 0 && t == 0) db[co] = bacc;
    }
}

// ---- BatchNorm (training mode) -----------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per channel: batch mean / biased variance over N = B*64, running statistics updated
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, float* __restrict__ mean, float* __restrict__ invstd,
                                                       float* __restrict__ rm, float* __restrict__ rv, int N, int C)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) s += y[(long)i * C + c];
    const float m = block_sum(s, red) / (float)N;
    float q = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) { const float d = y[(long)i * C + c] - m; q = fmaf(d, d, q); }
    const float var = block_sum(q, red) / (float)N;
    if (threadIdx.x == 0) {
        mean[c] = m;
        invstd[c] = 1.0f / sqrtf(var + BN_EPS);
        rm[c] = (1.0f - BN_MOMENTUM) * rm[c] + BN_MOMENTUM * m;
        rv[c] = (1.0f - BN_MOMENTUM) * rv[c] + BN_MOMENTUM * var * ((float)N / (float)(N - 1));
    }
}

// out = (skip ? skip : 0) + relu(gamma * (y - mean) * invstd + beta)
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ g,
                                                          const float* __restrict__ be, const float* __restrict__ skip,
                                                          float* __restrict__ out, long total, int C)
{
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        float z = fmaf(g[c], (y[idx] - mean[c]) * invstd[c], be[c]);
        z = z < 0.0f ? 0.0f : z;
        out[idx] = skip ? skip[idx] + z : z;
    }
}

// one workgroup per channel: dz = dout * (z > 0); dgamma = sum dz * xhat, dbeta = sum dz
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ g, const float* __restrict__ be,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    float sg = 0.0f, sb = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float xh = (y[(long)i * C + c] - mean[c]) * invstd[c];
        const float z = fmaf(g[c], xh, be[c]);
        const float dz = z > 0.0f ? dout[(long)i * C + c] : 0.0f;
        sg = fmaf(dz, xh, sg);
        sb += dz;
    }
    sg = block_sum(sg, red);
    sb = block_sum(sb, red);
    if (threadIdx.x == 0) { dgamma[c] = sg; dbeta[c] = sb; }
}

// dy = gamma * invstd / N * (N * dz - dbeta - xhat * dgamma)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ g, const float* __restrict__ be,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           float* __restrict__ dy, long total, int C, int N)
{
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const float xh = (y[idx] - mean[c]) * invstd[c];
        const float z = fmaf(g[c], xh, be[c]);
        const float dz = z > 0.0f ? dout[idx] : 0.0f;
        dy[idx] = g[c] * invstd[c] / (float)N * ((float)N * dz - dbeta[c] - xh * dgamma[c]);
    }
}

// ---- policy head tail: softmax over 4672 + policy loss and its gradient (nn.cpp:78-80, 99-103) ------
// one workgroup per board; loss[b] = -sum obs_p * log(p + 0.001);  dlogit_i = p_i * (g_i - sum_j g_j p_j),
// g_i = -obs_p_i / (p_i + 0.001)
__global__ __launch_bounds__(256) void policy_loss_kernel(const float* __restrict__ logits, const float* __restrict__ obsp,
                                                          float* __restrict__ dlogits, float* __restrict__ loss_rows)
{
    __shared__ float red[4];
    __shared__ float bc;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (size_t)b * KH_PSIZE;
    const float* t = obsp + (size_t)b * KH_PSIZE;
    float m = -INFINITY;
    for (int i = tid; i < KH_PSIZE; i += 256) m = fmaxf(m, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.0f;
    for (int i = tid; i < KH_PSIZE; i += 256) s += expf(x[i] - m);
    s = block_sum(s, red);
    const float ls = logf(s);
    float loss = 0.0f, gp = 0.0f;
    for (int i = tid; i < KH_PSIZE; i += 256) {
        const float p = expf((x[i] - m) - ls);
        loss -= t[i] * logf(p + 0.001f);
        gp += -t[i] / (p + 0.001f) * p;
    }
    loss = block_sum(loss, red);
    gp = block_sum(gp, red);
    if (tid == 0) { loss_rows[b] = loss; bc = gp; }
    __syncthreads();
    gp = bc;
    for (int i = tid; i < KH_PSIZE; i += 256) {
        const float p = expf((x[i] - m) - ls);
        dlogits[(size_t)b * KH_PSIZE + i] = p * (-t[i] / (p + 0.001f) - gp);
    }
}

// ---- value head tail: Linear(64,256) + tanh, MSE against the broadcast target (nn.cpp:86-88, 96) ------
// one workgroup per board, thread j = output j: v = tanh(h . W[j] + b[j]);
// dpre[b][j] = 2 (v - obs_v[b]) / (B*256) * (1 - v^2);  sq[b] = sum_j (v - obs_v[b])^2
__global__ __launch_bounds__(256) void value_fwd_loss_kernel(const float* __restrict__ h, const float* __restrict__ fcw,
                                                             const float* __restrict__ fcb, const float* __restrict__ obsv,
                                                             float* __restrict__ dpre, float* __restrict__ sq_rows, int B)
{
    __shared__ float hs[64];
    __shared__ float red[4];
    const int b = blockIdx.x, j = threadIdx.x;
    if (j < 64) hs[j] = h[(size_t)b * 64 + j];
    __syncthreads();
    float acc = fcb[j];
    for (int k = 0; k < 64; ++k) acc = fmaf(hs[k], fcw[(size_t)j * 64 + k], acc);
    const float v = tanhf(acc), d = v - obsv[b];
    dpre[(size_t)b * 256 + j] = 2.0f * d / (float)(B * 256) * (1.0f - v * v);
    const float sq = block_sum(d * d, red);
    if (j == 0) sq_rows[b] = sq;
}

// dW[j][k] = sum_b dpre[b][j] h[b][k]; db[j] = sum_b dpre[b][j]   (one thread per (j,k))
__global__ __launch_bounds__(256) void fc_wgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ h,
                                                       float* __restrict__ dw, float* __restrict__ db, int B)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 256 * 64) return;
    const int j = idx / 64, k = idx % 64;
    float acc = 0.0f, bacc = 0.0f;
    for (int b = 0; b < B; ++b) { const float g = dpre[(size_t)b * 256 + j]; acc = fmaf(g, h[(size_t)b * 64 + k], acc); bacc += g; }
    dw[idx] = acc;
    if (k == 0) db[j] = bacc;
}

// dh[b][k] = sum_j dpre[b][j] W[j][k]
__global__ __launch_bounds__(64) void fc_dgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ fcw,
                                                      float* __restrict__ dh)
{
    const int b = blockIdx.x, k = threadIdx.x;
    float acc = 0.0f;
    for (int j = 0; j < 256; ++j) acc = fmaf(dpre[(size_t)b * 256 + j], fcw[(size_t)j * 64 + k], acc);
    dh[(size_t)b * 64 + k] = acc;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float lr, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = fmaf(-lr, g[i], p[i]);
}

inline int nblocks(long total) { long b = (total + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

struct ConvBNOff { size_t w, b, g, be, rm, rv; int Ci, Co, T; };

}  // namespace

struct TrainNet {
    int F, C, R;
    ConvBNOff stem, pconv, vconv;
    std::vector<ConvBNOff> res;
    size_t p2w, p2b, fcw, fcb, total;
};

static TrainNet layout(int F, int C, int R)
{
    TrainNet n;
    n.F = F; n.C = C; n.R = R;
    size_t o = 0;
    auto convbn = [&](int Ci, int Co, int T) {
        ConvBNOff c;
        c.Ci = Ci; c.Co = Co; c.T = T;
        c.w = o; o += (size_t)Co * Ci * T;
        c.b = o; o += Co; c.g = o; o += Co; c.be = o; o += Co; c.rm = o; o += Co; c.rv = o; o += Co;
        return c;
    };
    n.stem = convbn(F, C, 9);
    for (int i = 0; i < 2 * R; ++i) n.res.push_back(convbn(C, C, 9));
    n.pconv = convbn(C, KH_POLICY_MID, 1);
    n.p2w = o; o += (size_t)KH_POLICY_PLANES * KH_POLICY_MID;
    n.p2b = o; o += KH_POLICY_PLANES;
    n.vconv = convbn(C, 1, 1);
    n.fcw = o; o += (size_t)KH_VALUE_WIDTH * 64;
    n.fcb = o; o += KH_VALUE_WIDTH;
    n.total = o;
    return n;
}

// One SGD step on device buffers (StepBuffers: params = blob updated in place, grads = blob-shaped,
// work = train_workspace_floats() floats of activations, conv outputs, statistics and gradients).

size_t train_workspace_floats(int F, int C, int R, int B)
{
    const size_t act = (size_t)B * 64 * (size_t)(C > KH_POLICY_MID ? C : KH_POLICY_MID);
    const int L = 1 + 2 * R;
    // per tower layer: output activation + conv output; heads: pconv y/act, logits, dlogits, value pieces; 3 gradient planes
    return act * (2 * (size_t)L + 12) + (size_t)B * 64 * F + (size_t)B * KH_PSIZE * 2 + (size_t)(2 * L + 8) * 2 * 256 + 65536;
}

hipError_t train_step(const TrainNet& n, const StepBuffers& sb, const float* x_in, const float* obsp, const float* obsv,
                      int B, float lr, float* loss_rows /* [2*B] device: policy rows, value squared-error rows */, hipStream_t s)
{
    const int C = n.C, N = B * 64;
    float* P = sb.params;
    float* G = sb.grads;
    float* wk = sb.work;
    auto take = [&](size_t nfl) { float* r = wk; wk += nfl; return r; };
    const size_t actC = (size_t)B * 64 * C;
    struct Saved { const float* in; float* y; float* out; float* mean; float* invstd; };
    std::vector<Saved> sv;
    (void)hipMemsetAsync(G, 0, n.total * sizeof(float), s);

    auto fwd = [&](const ConvBNOff& c, const float* in, const float* skip) {
        Saved v;
        v.in = in;
        v.y = take((size_t)N * c.Co); v.out = take((size_t)N * c.Co); v.mean = take(256); v.invstd = take(256);
        hipLaunchKernelGGL(conv_fwd_kernel, dim3(nblocks((long)N * c.Co)), dim3(256), 0, s, P + c.w, P + c.b, in, v.y, B, c.Ci, c.Co, c.T);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(c.Co), dim3(256), 0, s, v.y, v.mean, v.invstd, P + c.rm, P + c.rv, N, c.Co);
        hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(nblocks((long)N * c.Co)), dim3(256), 0, s, v.y, v.mean, v.invstd, P + c.g, P + c.be, skip, v.out,
                           (long)N * c.Co, c.Co);
        sv.push_back(v);
        return v.out;
    };
    // dout: gradient w.r.t. relu(bn(conv(in))) ; writes parameter gradients, returns dL/d(in) in `din` (accumulated if acc)
    auto bwd = [&](const ConvBNOff& c, const Saved& v, const float* dout, float* dy_tmp, float* din, int acc) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(c.Co), dim3(256), 0, s, dout, v.y, v.mean, v.invstd, P + c.g, P + c.be, G + c.g, G + c.be, N, c.Co);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblocks((long)N * c.Co)), dim3(256), 0, s, dout, v.y, v.mean, v.invstd, P + c.g, P + c.be,
                           G + c.g, G + c.be, dy_tmp, (long)N * c.Co, c.Co, N);
        hipLaunchKernelGGL(conv_wgrad_kernel, dim3(nblocks((long)c.Co * c.Ci * c.T)), dim3(256), 0, s, dy_tmp, v.in, G + c.w, G + c.b, B, c.Ci, c.Co, c.T);
        if (din)
            hipLaunchKernelGGL(conv_dgrad_kernel, dim3(nblocks((long)N * c.Ci)), dim3(256), 0, s, P + c.w, dy_tmp, din, B, c.Ci, c.Co, c.T, acc);
    };

    // ---- forward (nn.cpp:59-91, module in train mode) ----
    const float* x = fwd(n.stem, x_in, nullptr);
    for (int r = 0; r < n.R; ++r) {
        const float* t = fwd(n.res[2 * r], x, nullptr);
        x = fwd(n.res[2 * r + 1], t, x);                    // x + relu(bn2(conv2(...)))   nn.cpp:26-34
    }
    const size_t tower_saved = sv.size();
    const float* pm = fwd(n.pconv, x, nullptr);             // [B][64][128]
    float* logits = take((size_t)B * KH_PSIZE);
    hipLaunchKernelGGL(conv_fwd_kernel, dim3(nblocks((long)N * KH_POLICY_PLANES)), dim3(256), 0, s, P + n.p2w, P + n.p2b, pm, logits, B,
                       KH_POLICY_MID, KH_POLICY_PLANES, 1);     // [B][64][73] = index pixel*73 + plane (nn.cpp:78-79)
    const float* h = fwd(n.vconv, x, nullptr);              // [B][64][1]

    // ---- losses and their gradients ----
    float* dlogits = take((size_t)B * KH_PSIZE);
    hipLaunchKernelGGL(policy_loss_kernel, dim3(B), dim3(256), 0, s, logits, obsp, dlogits, loss_rows);
    float* dpre = take((size_t)B * 256);
    hipLaunchKernelGGL(value_fwd_loss_kernel, dim3(B), dim3(256), 0, s, h, P + n.fcw, P + n.fcb, obsv, dpre, loss_rows + B, B);

    // ---- backward ----
    float* dX = take(actC);             // gradient w.r.t. the tower output, then walked down the tower
    float* dT = take((size_t)N * (C > KH_POLICY_MID ? C : KH_POLICY_MID));
    float* dtmp = take((size_t)N * (C > KH_POLICY_MID ? C : KH_POLICY_MID));
    // value head: fc -> relu/bn/conv
    float* dh = take((size_t)B * 64);
    hipLaunchKernelGGL(fc_wgrad_kernel, dim3(64), dim3(256), 0, s, dpre, h, G + n.fcw, G + n.fcb, B);
    hipLaunchKernelGGL(fc_dgrad_kernel, dim3(B), dim3(64), 0, s, dpre, P + n.fcw, dh);
    bwd(n.vconv, sv[tower_saved + 1], dh, dtmp, dX, 0);
    // policy head: conv2 (plain) -> relu/bn/conv
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(nblocks((long)KH_POLICY_PLANES * KH_POLICY_MID)), dim3(256), 0, s, dlogits, pm, G + n.p2w, G + n.p2b, B,
                       KH_POLICY_MID, KH_POLICY_PLANES, 1);
    hipLaunchKernelGGL(conv_dgrad_kernel, dim3(nblocks((long)N * KH_POLICY_MID)), dim3(256), 0, s, P + n.p2w, dlogits, dT, B, KH_POLICY_MID,
                       KH_POLICY_PLANES, 1, 0);
    bwd(n.pconv, sv[tower_saved], dT, dtmp, dX, 1);
    // tower
    for (int r = n.R - 1; r >= 0; --r) {
        // out = xin + relu(bn2(conv2(t))):  d t = through conv2;  d xin = dX (skip) + through conv1
        bwd(n.res[2 * r + 1], sv[2 + 2 * r], dX, dtmp, dT, 0);       // dT = dL/dt
        bwd(n.res[2 * r], sv[1 + 2 * r], dT, dtmp, dX, 1);           // dX += dL/dxin via conv1
    }
    bwd(n.stem, sv[0], dX, dtmp, nullptr, 0);

    hipLaunchKernelGGL(sgd_kernel, dim3(nblocks((long)n.total)), dim3(256), 0, s, P, G, lr, (long)n.total);
    return hipGetLastError();
}

TrainNet* train_layout_new(int F, int C, int R) { return new TrainNet(layout(F, C, R)); }
void train_layout_free(TrainNet* n) { delete n; }

}  // namespace kh
