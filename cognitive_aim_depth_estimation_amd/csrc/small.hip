// Small fp32 kernels for the per-image MLPs and heads (reference
// src/model.py:32-53 ambient, :311 focal projection, :342-346 focal fusion,
// :482-519 EXIF prior, :908-941 fusion / decision / confidence heads).
// Batch-many-rows x few-hundred-columns: GEMV class, one wave per output.
#include "common.h"
#include "../../include/cogaim.h"
#include <dlfcn.h>

namespace {

// out[b*ldo + o] = act(x[b*ldx + :K] . W[o*K + :K] + bias[o] + bias2[o] * sum_{p < scale_parts} bias2_scale[b*scale_parts + p])
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ bias2, const float* __restrict__ bias2_scale,
                                                         int scale_parts, float* __restrict__ out, int ldo, int B, int O,
                                                         int K, int act) {
  const int lane = threadIdx.x & 63;
  const long long wid = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (wid >= (long long)B * O) return;
  const int b = (int)(wid / O), o = (int)(wid - (long long)b * O);
  const float* xr = x + (size_t)b * ldx;
  const float* wr = W + (size_t)o * K;
  float acc = 0.f;
  if (((K | ldx) & 3) == 0) {
    for (int k = lane * 4; k < K; k += 256) {
      const float4 a = *reinterpret_cast<const float4*>(xr + k);
      const float4 w = *reinterpret_cast<const float4*>(wr + k);
      acc += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
    }
  } else {
    for (int k = lane; k < K; k += 64) acc += xr[k] * wr[k];
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float y = acc + (bias ? bias[o] : 0.f);
    if (bias2) {
      float sc = 1.0f;
      if (bias2_scale) {
        sc = 0.f;
        for (int p = 0; p < scale_parts; ++p) sc += bias2_scale[(size_t)b * scale_parts + p];      // parts ascending: fixed order
      }
      y += bias2[o] * sc;
    }
    if (act == 1) y = y < 0.f ? 0.f : y;      // ReLU that keeps a NaN a NaN, as torch's does (fmaxf would drop it)
    out[(size_t)b * ldo + o] = y;
  }
}

// feat[b][0:E] = emb[idx[b]][0:E];  cont[b] = {focal_length, aperture, log(iso + 1)}
// An index outside [0, num_emb) raises *err (when given) and fills the image's row with NaN instead of
// faulting: depth and confidence of that image come out NaN, so the bad result cannot be mistaken for a
// prediction even by a caller that never looks at the flag.
__global__ void exif_prep_kernel(const long long* __restrict__ idx, const float* __restrict__ fl,
                                 const float* __restrict__ ap, const float* __restrict__ iso,
                                 const float* __restrict__ emb, int num_emb, int E, float* __restrict__ feat, int ldf,
                                 float* __restrict__ cont, int B, int* __restrict__ err) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (b >= B) return;
  long long row = idx[b];
  const bool bad = row < 0 || row >= num_emb;
  if (bad) {
    if (t == 0 && err) *err = 1;
    row = 0;
  }
  for (int e = t; e < E; e += blockDim.x)
    feat[(size_t)b * ldf + e] = bad ? __builtin_nanf("") : emb[(size_t)row * E + e];
  if (t == 0) { cont[b * 3] = fl[b]; cont[b * 3 + 1] = ap[b]; cont[b * 3 + 2] = logf(iso[b] + 1.0f); }
}

// depth = softplus(f . wd + bd) (beta 1, threshold 20); conf = sigmoid(w2 * relu(f . wc + bc) + b2)
__global__ __launch_bounds__(256) void heads_kernel(const float* __restrict__ fused, int K, const float* __restrict__ wd,
                                                    const float* __restrict__ bd, const float* __restrict__ wc,
                                                    const float* __restrict__ bc, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, float* __restrict__ depth,
                                                    float* __restrict__ conf, int B) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float ad = 0.f, ac = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float f = fused[(size_t)b * K + k];
    ad += f * wd[k]; ac += f * wc[k];
  }
  ad = wave_sum(ad); ac = wave_sum(ac);
  if (lane == 0) {
    const float zd = ad + bd[0];
    depth[b] = zd > 20.f ? zd : log1pf(expf(zd));
    const float hc = ac + bc[0];
    const float zc = (hc < 0.f ? 0.f : hc) * w2[0] + b2[0];      // NaN-keeping ReLU
    conf[b] = 1.0f / (1.0f + expf(-zc));
  }
}

// ---- several small layers in ONE launch ---------------------------------------------------------------------------
// One 1024-thread workgroup per image walks up to eight Linear layers whose activations stay in LDS (4096 floats):
//   out_l[o] = act(W_l[o, :K] . in_l + b_l[o] + b2_l[o] * scale[image])
// `in_l` / `out_l` are offsets into the LDS array, so a layer can read the concatenation of an input vector and an
// earlier layer's output (EXIF prior: embedding | encoder output).  The first input may be given as `parts` partial
// vectors that are summed while they are loaded (pairs of running sums, parts ascending: the order of
// reduce_partials_kernel).  A wave owns four outputs at a time; the lanes stride over K.
constexpr int MLP_LDS = 4096;
constexpr int MLP_MAX_LAYERS = 8;
struct MlpLayerDev { const float* W; const float* b; const float* b2; int K, O, act, in_off, out_off, pad; };
struct MlpChainDev {
  MlpLayerDev layer[MLP_MAX_LAYERS];
  int n_layers, in0_len, in0_off, in0_parts;
  const float* in0; long long in0_ld, in0_part_stride;
  const float* in1; long long in1_ld; int in1_len, in1_off;
  const float* scale;
  float* out; long long out_ld; int out_len, out_off;
};

// One Linear layer on vectors held in LDS by a 16-wave workgroup: a wave owns four outputs at a time (their weight rows'
// loads are issued together), the lanes stride over K.  `copy` (may be NULL) also receives the outputs.
__device__ __forceinline__ void mlp_layer(const float* in, float* out, const float* __restrict__ W, const float* __restrict__ b,
                                          const float* __restrict__ b2, float sc, int K, int O, int act, bool in_aligned,
                                          int wave, int lane, float* __restrict__ copy) {
  const bool vec = (K & 3) == 0 && in_aligned && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
  if (vec && K <= 256) {
    // narrow layer: the lane's one chunk of every weight row of up to four passes (16 outputs) is requested before the
    // first reduction, so that a layer costs ONE memory round trip however many passes it takes
#pragma unroll 1
    for (int o0 = wave * 4; o0 < O; o0 += 256) {
      const int k = lane * 4;
      const bool on = k < K;
      const float4 a = on ? *reinterpret_cast<const float4*>(in + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 ww[4][4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          ww[ps][j] = *reinterpret_cast<const float4*>(W + (size_t)min(o0 + 64 * ps + j, O - 1) * K + (on ? k : 0));
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        if (o0 + 64 * ps >= O) break;
        float acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = wave_sum(on ? (a.x * ww[ps][j].x + a.y * ww[ps][j].y) + (a.z * ww[ps][j].z + a.w * ww[ps][j].w) : 0.f);
        const int o = o0 + 64 * ps + lane;
        if (lane < 4 && o < O) {
          float y = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
          if (b) y += b[o];
          if (b2) y += b2[o] * sc;
          if (act == 1) y = y < 0.f ? 0.f : y;
          out[o] = y;
          if (copy) copy[o] = y;
        }
      }
    }
    return;
  }
#pragma unroll 1
  for (int o0 = wave * 4; o0 < O; o0 += 64) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = W + (size_t)min(o0 + j, O - 1) * K;
    if (vec) {
      for (int k = lane * 4; k < K; k += 256) {
        const float4 a = *reinterpret_cast<const float4*>(in + k);
        float4 ww[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ww[j] = *reinterpret_cast<const float4*>(w[j] + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += (a.x * ww[j].x + a.y * ww[j].y) + (a.z * ww[j].z + a.w * ww[j].w);
      }
    } else {
      for (int k = lane; k < K; k += 64) {
        const float a = in[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += a * w[j][k];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = wave_sum(acc[j]);
    if (lane < 4 && o0 + lane < O) {
      const int o = o0 + lane;
      float y = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
      if (b) y += b[o];
      if (b2) y += b2[o] * sc;
      if (act == 1) y = y < 0.f ? 0.f : y;          // ReLU that keeps a NaN a NaN, as torch's does
      out[o] = y;
      if (copy) copy[o] = y;
    }
  }
}

__global__ __launch_bounds__(1024) void mlp_chain_kernel(MlpChainDev d) {
  __shared__ __attribute__((aligned(16))) float buf[MLP_LDS];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const float* p = d.in0 + (size_t)img * d.in0_ld;
    for (int i = tid; i < d.in0_len; i += 1024) {
      float s0 = 0.f, s1 = 0.f;
      int k = 0;
      for (; k + 1 < d.in0_parts; k += 2) { s0 += p[(size_t)k * d.in0_part_stride + i]; s1 += p[(size_t)(k + 1) * d.in0_part_stride + i]; }
      if (k < d.in0_parts) s0 += p[(size_t)k * d.in0_part_stride + i];
      buf[d.in0_off + i] = d.in0_parts == 1 ? p[i] : s0 + s1;
    }
    if (d.in1)
      for (int i = tid; i < d.in1_len; i += 1024) buf[d.in1_off + i] = d.in1[(size_t)img * d.in1_ld + i];
  }
  __syncthreads();
  const float sc = d.scale ? d.scale[img] : 1.0f;
#pragma unroll 1
  for (int l = 0; l < d.n_layers; ++l) {
    const MlpLayerDev L = d.layer[l];
    mlp_layer(buf + L.in_off, buf + L.out_off, L.W, L.b, L.b2, sc, L.K, L.O, L.act, (L.in_off & 3) == 0, wave, lane, nullptr);
    __syncthreads();
  }
  for (int i = tid; i < d.out_len; i += 1024) d.out[(size_t)img * d.out_ld + i] = buf[d.out_off + i];
}

// fused[b] = relu(Wf cat[b] + bf) (src/model.py:908-912, 1045), then the two heads (:925-941, 1120-1123) from it:
// one workgroup per image, the fused row stays in LDS.
constexpr int FH_MAX = 1024;
__global__ __launch_bounds__(1024) void fusion_heads_kernel(const float* __restrict__ cat, int ldcat, int K,
                                                            const float* __restrict__ wf, const float* __restrict__ bf, int F,
                                                            const float* __restrict__ wd, const float* __restrict__ bd,
                                                            const float* __restrict__ wc, const float* __restrict__ bc,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            float* __restrict__ fused, float* __restrict__ depth,
                                                            float* __restrict__ conf) {
  __shared__ __attribute__((aligned(16))) float xin[FH_MAX], f[FH_MAX];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < K; i += 1024) xin[i] = cat[(size_t)b * ldcat + i];
  __syncthreads();
  mlp_layer(xin, f, wf, bf, nullptr, 1.0f, K, F, 1, true, wave, lane, fused + (size_t)b * F);
  __syncthreads();
  if (wave < 2) {
    const float* w = wave == 0 ? wd : wc;
    float acc = 0.f;
    for (int k = lane; k < F; k += 64) acc += f[k] * w[k];
    acc = wave_sum(acc);
    if (lane == 0) {
      if (wave == 0) {
        const float zd = acc + bd[0];
        depth[b] = zd > 20.f ? zd : log1pf(expf(zd));
      } else {
        const float hc = acc + bc[0];
        const float zc = (hc < 0.f ? 0.f : hc) * w2[0] + b2[0];      // NaN-keeping ReLU
        conf[b] = 1.0f / (1.0f + expf(-zc));
      }
    }
  }
}

// Results of one forward out of the (fixed) workspace into the caller's fresh tensors, one launch: depth, confidence,
// fused features, attention rows, and the device-side condition words into a host-mapped snapshot.
__global__ __launch_bounds__(256) void export_outputs_kernel(const float* __restrict__ depth_src, const float* __restrict__ conf_src,
                                                             const float* __restrict__ fused_src, int F,
                                                             const float* __restrict__ attn_src, int N,
                                                             float* __restrict__ depth, float* __restrict__ conf,
                                                             float* __restrict__ fused, float* __restrict__ attn,
                                                             const int* __restrict__ flags, int* __restrict__ flags_host,
                                                             int n_flags, int B) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b == B) {
    if (flags_host)
      for (int i = tid; i < n_flags; i += 256) flags_host[i] = flags[i];
    return;
  }
  if (tid == 0) { depth[b] = depth_src[b]; conf[b] = conf_src[b]; }
  if (fused) for (int i = tid; i < F; i += 256) fused[(size_t)b * F + i] = fused_src[(size_t)b * F + i];
  if (attn) for (int i = tid; i < N; i += 256) attn[(size_t)b * N + i] = attn_src[(size_t)b * N + i];
}


// Heat map for display (reference demo.py:533-552), one workgroup per image: cube the attention weights, multiply
// everything at or below their 70th percentile by 0.3, rescale to [0, 1], pad to the G x G grid with zeros.
// The percentile is numpy's default ("linear"): the weights are sorted in LDS (bitonic, padded with +inf to a power
// of two) and the two order statistics around 0.7 (N - 1) interpolated the way numpy's _lerp does.
constexpr int HM_MAX = 6144;      // weights per image (74 x 74 = 5476 at 1036 pixels); sort buffer: next power of two
__global__ __launch_bounds__(256) void heatmap_kernel(const float* __restrict__ attn, float* __restrict__ out, int N,
                                                      int GG, int npad) {
  __shared__ float val[HM_MAX], srt[8192];
  __shared__ float red[4];
  __shared__ float thr_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* ab = attn + (size_t)b * N;
  for (int i = tid; i < npad; i += 256) {
    float v = INFINITY;
    if (i < N) { const float a = ab[i]; v = a * a * a; val[i] = v; }
    srt[i] = v;
  }
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const float x = srt[i], y = srt[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { srt[i] = y; srt[l] = x; }
        }
      }
      __syncthreads();
    }
  if (tid == 0) {
    const double pos = 0.7 * (double)(N - 1);
    const int lo = (int)pos, hi = lo + 1 < N ? lo + 1 : N - 1;
    const float t = (float)(pos - (double)lo);
    const float a = srt[lo], c = srt[hi], diff = c - a;
    thr_s = t >= 0.5f ? c - diff * (1.0f - t) : a + diff * t;
  }
  __syncthreads();
  const float thr = thr_s;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = tid; i < N; i += 256) {
    float v = val[i];
    v = v > thr ? v : v * 0.3f;
    val[i] = v; mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  mx = block_max_256(mx, red);
  mn = -block_max_256(-mn, red);
  const float inv = 1.0f / (mx - mn + 1e-8f);
  float* ob = out + (size_t)b * GG;
  for (int i = tid; i < GG; i += 256) ob[i] = i < N ? (val[i] - mn) * inv : 0.f;
}

// Curiosity module, element-wise parts (reference src/model.py:608-610, 743-745):
//   z = mu + eps * exp(0.5 * logvar)   [B, L];   noisy = cls + noise   [B, D]   (cls rows ld_cls apart)
__global__ __launch_bounds__(256) void curiosity_sample_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                               const float* __restrict__ eps, float* __restrict__ z,
                                                               const float* __restrict__ cls, long long ld_cls,
                                                               const float* __restrict__ noise, float* __restrict__ noisy,
                                                               int L, int D) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < L; i += 256)
    z[(size_t)b * L + i] = mu[(size_t)b * L + i] + eps[(size_t)b * L + i] * expf(0.5f * logvar[(size_t)b * L + i]);
  for (int i = threadIdx.x; i < D; i += 256) noisy[(size_t)b * D + i] = cls[b * ld_cls + i] + noise[(size_t)b * D + i];
}

// Curiosity reward per image (src/model.py:616-686 with loss_type "robust", uncertainty_weight 0.1, no EXIF:
// geometric term 0.5, :698-700) and, by one thread after all images, the ring-buffer update of :760-773 in
// image order.  One workgroup; images are walked by the waves.
__global__ __launch_bounds__(256) void curiosity_finish_kernel(const float* __restrict__ rec, const float* __restrict__ cls,
                                                               long long ld_cls, const float* __restrict__ mu,
                                                               const float* __restrict__ logvar,
                                                               const float* __restrict__ unc_pre,
                                                               const float* __restrict__ lb_pre,
                                                               const float* __restrict__ ln_pre,
                                                               const float* __restrict__ weights3, float* __restrict__ reward,
                                                               float* __restrict__ history, long long* __restrict__ pointer,
                                                               int B, int L, int hist_len) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float w0 = weights3[0], w1 = weights3[1], w2 = weights3[2];
  const float wm = fmaxf(w0, fmaxf(w1, w2));
  const float e0 = expf(w0 - wm), e1 = expf(w1 - wm), e2 = expf(w2 - wm), es = e0 + e1 + e2;
  for (int b = wave; b < B; b += 4) {
    float sq = 0.f, kl = 0.f;
    for (int i = lane; i < L; i += 64) {
      const float d = rec[(size_t)b * L + i] - cls[b * ld_cls + i];
      const float m = mu[(size_t)b * L + i], lv = logvar[(size_t)b * L + i];
      sq += d * d;
      kl += 1.0f + lv - m * m - expf(lv);
    }
    sq = wave_sum(sq); kl = wave_sum(kl);
    if (lane == 0) {
      float re = sqrtf(sq + 1e-8f);
      re = fmaxf(re / (1.0f + re), 0.f);
      const float kld = fmaxf(-0.5f * kl, 0.f);
      const float up = unc_pre[b];
      const float unc = fminf(fmaxf(up > 20.f ? up : log1pf(expf(up)), 0.f), 10.f);
      const float basic = re + 0.1f * kld + 0.1f * unc;
      const float lb = 1.0f / (1.0f + expf(-lb_pre[b])), ln = 1.0f / (1.0f + expf(-ln_pre[b]));
      const float local = fminf(fmaxf(lb + fabsf(lb - ln) * 0.2f, 0.f), 1.f);
      reward[b] = (e0 / es) * 0.5f + (e1 / es) * local + (e2 / es) * basic;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && history && pointer) {
    long long p = *pointer;
    for (int b = 0; b < B; ++b) {
      const long long idx = ((p % hist_len) + hist_len) % hist_len;
      history[idx] = reward[b];
      p = (p + 1) % hist_len;
    }
    *pointer = p;
  }
}

}  // namespace

extern "C" {

// src/model.py:608-610 and :743-745 (see curiosity_sample_kernel)
int cogaim_curiosity_sample(const float* mu, const float* logvar, const float* eps, float* z, const float* cls,
                            long long ld_cls, const float* noise, float* noisy, int B, int L, int D, void* stream) {
  if (B <= 0 || L <= 0 || D <= 0) return COGAIM_EINVAL;
  hipLaunchKernelGGL(curiosity_sample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, z, cls, ld_cls,
                     noise, noisy, L, D);
  return COGAIM_LAUNCH_CHECK();
}

// src/model.py:616-686 + :760-773 (see curiosity_finish_kernel); history / pointer may be NULL (no ring update)
int cogaim_curiosity_finish(const float* rec, const float* cls, long long ld_cls, const float* mu, const float* logvar,
                            const float* unc_pre, const float* lb_pre, const float* ln_pre, const float* weights3,
                            float* reward, float* history, long long* pointer, int B, int L, int hist_len, void* stream) {
  if (B <= 0 || L <= 0 || hist_len <= 0) return COGAIM_EINVAL;
  hipLaunchKernelGGL(curiosity_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rec, cls, ld_cls, mu, logvar,
                     unc_pre, lb_pre, ln_pre, weights3, reward, history, pointer, B, L, hist_len);
  return COGAIM_LAUNCH_CHECK();
}


// act: 0 identity, 1 ReLU.  bias / bias2 / bias2_scale may be NULL.
int cogaim_linear_f32(const float* x, int ldx, const float* W, const float* bias, const float* bias2,
                      const float* bias2_scale, int scale_parts, float* out, int ldo, int B, int O, int K, int act,
                      void* stream) {
  if (B <= 0 || O <= 0 || K <= 0 || ldx < K || (act != 0 && act != 1) || (bias2_scale && scale_parts < 1)) return COGAIM_EINVAL;
  const long long waves = (long long)B * O;
  hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, W,
                     bias, bias2, bias2_scale, scale_parts, out, ldo, B, O, K, act);
  return COGAIM_LAUNCH_CHECK();
}

// Up to eight small Linear layers per image in one launch (see mlp_chain_kernel).  `desc` is read on the host.
int cogaim_mlp_chain(const cogaim_mlp_chain_desc* desc, int B, void* stream) {
  if (!desc || B <= 0 || desc->n_layers < 1 || desc->n_layers > MLP_MAX_LAYERS || !desc->in0 || !desc->out) return COGAIM_EINVAL;
  auto region_ok = [](long long off, long long len) { return off >= 0 && len >= 0 && off + len <= MLP_LDS; };
  if (!region_ok(desc->in0_off, desc->in0_len) || desc->in0_len < 1 || desc->in0_parts < 1 ||
      (desc->in1 && !region_ok(desc->in1_off, desc->in1_len)) || !region_ok(desc->out_off, desc->out_len) ||
      desc->out_len < 1 || desc->out_ld < desc->out_len || desc->in0_ld < 0)
    return COGAIM_EINVAL;
  MlpChainDev d{};
  for (int l = 0; l < desc->n_layers; ++l) {
    const cogaim_mlp_layer& L = desc->layer[l];
    if (!L.W || L.K < 1 || L.O < 1 || (L.act != 0 && L.act != 1) || !region_ok(L.in_off, L.K) || !region_ok(L.out_off, L.O))
      return COGAIM_EINVAL;
    // a layer must not overwrite what it is still reading
    if (L.out_off < L.in_off + L.K && L.in_off < L.out_off + L.O) return COGAIM_EINVAL;
    d.layer[l] = MlpLayerDev{L.W, L.b, L.b2, L.K, L.O, L.act, L.in_off, L.out_off, 0};
  }
  d.n_layers = desc->n_layers;
  d.in0 = desc->in0; d.in0_ld = desc->in0_ld; d.in0_part_stride = desc->in0_part_stride;
  d.in0_len = desc->in0_len; d.in0_off = desc->in0_off; d.in0_parts = desc->in0_parts;
  d.in1 = desc->in1; d.in1_ld = desc->in1_ld; d.in1_len = desc->in1 ? desc->in1_len : 0; d.in1_off = desc->in1_off;
  d.scale = desc->scale;
  d.out = desc->out; d.out_ld = desc->out_ld; d.out_len = desc->out_len; d.out_off = desc->out_off;
  hipLaunchKernelGGL(mlp_chain_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, d);
  return COGAIM_LAUNCH_CHECK();
}

// fused [B,F] = relu(Wf cat + bf); depth[b] = softplus(wd . fused + bd); conf[b] = sigmoid(w2 relu(wc . fused + bc) + b2)
int cogaim_fusion_heads(const float* cat, int ldcat, int K, const float* wf, const float* bf, int F, const float* wd,
                        const float* bd, const float* wc, const float* bc, const float* w2, const float* b2, float* fused,
                        float* depth, float* conf, int B, void* stream) {
  if (B <= 0 || K <= 0 || K > FH_MAX || F <= 0 || F > FH_MAX || ldcat < K || !cat || !wf || !bf || !fused || !depth || !conf)
    return COGAIM_EINVAL;
  hipLaunchKernelGGL(fusion_heads_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, cat, ldcat, K, wf, bf, F, wd, bd, wc,
                     bc, w2, b2, fused, depth, conf);
  return COGAIM_LAUNCH_CHECK();
}

// 0 when a kernel may store through `host_ptr` (page-locked, mapped host memory whose device address equals its host
// address), else -22.  Lets a caller decide once whether cogaim_export_outputs may be given a host snapshot buffer.
int cogaim_host_pointer_ok(void* host_ptr) {
  if (!host_ptr) return COGAIM_EINVAL;
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, host_ptr, 0) != hipSuccess) { (void)hipGetLastError(); return COGAIM_EINVAL; }
  return dev == host_ptr ? COGAIM_OK : COGAIM_EINVAL;
}

// One launch that copies a forward's results out of fixed workspace buffers: depth / conf [B], fused [B,F] (dst NULL:
// skipped), attention [B,N] (dst NULL: skipped) and n_flags condition words into flags_host (NULL: skipped; must have
// passed cogaim_host_pointer_ok).
int cogaim_export_outputs(const float* depth_src, const float* conf_src, const float* fused_src, int F,
                          const float* attn_src, int N, float* depth, float* conf, float* fused, float* attn,
                          const int* flags, int* flags_host, int n_flags, int B, void* stream) {
  if (B <= 0 || !depth_src || !conf_src || !depth || !conf || (fused && (!fused_src || F <= 0)) ||
      (attn && (!attn_src || N <= 0)) || (flags_host && (!flags || n_flags <= 0)))
    return COGAIM_EINVAL;
  hipLaunchKernelGGL(export_outputs_kernel, dim3(B + 1), dim3(256), 0, (hipStream_t)stream, depth_src, conf_src, fused_src,
                     F, attn_src, N, depth, conf, fused, attn, flags, flags_host, n_flags, B);
  return COGAIM_LAUNCH_CHECK();
}

int cogaim_exif_prep(const long long* camera_idx, const float* focal_length, const float* aperture, const float* iso,
                     const float* embedding, int num_embeddings, int emb_dim, float* feat, int ldf, float* cont, int B,
                     int* err_flag, void* stream) {
  if (B <= 0 || emb_dim <= 0 || num_embeddings <= 0 || ldf < emb_dim) return COGAIM_EINVAL;
  hipLaunchKernelGGL(exif_prep_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, camera_idx, focal_length, aperture,
                     iso, embedding, num_embeddings, emb_dim, feat, ldf, cont, B, err_flag);
  return COGAIM_LAUNCH_CHECK();
}

int cogaim_heads(const float* fused, int K, const float* wd, const float* bd, const float* wc, const float* bc,
                 const float* w2, const float* b2, float* depth, float* conf, int B, void* stream) {
  if (B <= 0 || K <= 0) return COGAIM_EINVAL;
  hipLaunchKernelGGL(heads_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, fused, K, wd, bd, wc, bc, w2,
                     b2, depth, conf, B);
  return COGAIM_LAUNCH_CHECK();
}

// out[b][G*G] = display heat map of attn[b][N] (G = ceil(sqrt(N)); N <= 6144).
int cogaim_heatmap_postprocess(const float* attn, float* out, int B, int N, int G, void* stream) {
  if (B <= 0 || N <= 0 || N > HM_MAX || G <= 0 || (long long)G * G < N || (long long)(G - 1) * (G - 1) >= N)
    return COGAIM_EINVAL;
  int npad = 1;
  while (npad < N) npad <<= 1;
  hipLaunchKernelGGL(heatmap_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, attn, out, N, G * G, npad);
  return COGAIM_LAUNCH_CHECK();
}

// All-gather of per-image result rows over RCCL (xGMI inside a node): every rank contributes `count` floats, recvbuf
// receives world * count in rank order.  `comm` is the caller's ncclComm_t.  RCCL is not a link-time dependency of this
// library: the entry point is looked up among the libraries the process has already loaded (torch ships one), so a
// process without RCCL gets -38 instead of a load error.
int cogaim_allgather_f32(const float* sendbuf, float* recvbuf, long long count, void* comm, void* stream) {
  if (!sendbuf || !recvbuf || count <= 0 || !comm) return COGAIM_EINVAL;
  using AllGatherFn = int (*)(const void*, void*, size_t, int, void*, hipStream_t);      // ncclAllGather
  static AllGatherFn fn = nullptr;                       // looked up again on every call until a loaded RCCL is found
  if (!fn) fn = reinterpret_cast<AllGatherFn>(dlsym(RTLD_DEFAULT, "ncclAllGather"));
  if (!fn) return -38;
  constexpr int kNcclFloat32 = 7;                                                         // ncclFloat in rccl.h
  return fn(sendbuf, recvbuf, (size_t)count, kNcclFloat32, comm, (hipStream_t)stream) == 0 ? COGAIM_OK : -1;
}

int cogaim_version() { return 4; }

}  // extern "C"
