// Small fp32 kernels for the per-image MLPs and heads (reference
// src/model.py:32-53 ambient, :311 focal projection, :342-346 focal fusion,
// :482-519 EXIF prior, :908-941 fusion / decision / confidence heads).
// Batch-many-rows x few-hundred-columns: GEMV class, one wave per output.
#include "common.h"

namespace {

// out[b*ldo + o] = act(x[b*ldx + :K] . W[o*K + :K] + bias[o] * (bias_scale ? bias_scale[b] : 1))
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ bias_scale, float* __restrict__ out,
                                                         int ldo, int B, int O, int K, int act) {
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
    float bv = bias ? bias[o] : 0.f;
    if (bias_scale) bv *= bias_scale[b];
    float y = acc + bv;
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


// act: 0 identity, 1 ReLU.  bias / bias_scale may be NULL.
int cogaim_linear_f32(const float* x, int ldx, const float* W, const float* bias, const float* bias_scale, float* out,
                      int ldo, int B, int O, int K, int act, void* stream) {
  if (B <= 0 || O <= 0 || K <= 0 || ldx < K || (act != 0 && act != 1)) return COGAIM_EINVAL;
  const long long waves = (long long)B * O;
  hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, W,
                     bias, bias_scale, out, ldo, B, O, K, act);
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

int cogaim_version() { return 3; }

}  // extern "C"
