// Fused RAdam + L2-regularisation update over the flat parameter buffer (one launch for all ~90 variables), gfx950.
//
// Replaces utils/radam.py:56-107 (RAdam.get_updates: one TF op chain per variable) and the Keras-added L2 regularisers
// (backbone/basic_backbone.py:41,64,76: loss += lambda * sum(w^2), gradient 2*lambda*w).  The kernel reads p, g, m, v and
// writes p, m, v plus the bf16 compute copy of p (the operand of the next step's convolutions) = 30 B/param; it also zeroes
// g for the next step's atomically accumulated weight gradients and emits block partial sums of lambda*p^2 (the
// regularisation part of the reported loss, evaluated at the pre-update weights like Keras does).
#include "common.h"

namespace {

// schedule state in device memory (graph-capturable: no host scalar changes between replays)
//   f[0] = lr (set by the host when the epoch schedule changes), f[1] = lr_t, f[2] = rho_t,
//   f[3] = update rule of this step: 0 plain first moment (RAdam warm-up), 1 adaptive (RAdam rho_t >= 5, Adam), 2 SGD momentum + Nesterov, 3 SGD momentum
//   it[0] = iterations (int64)
// kind: 0 = RAdam (utils/radam.py:56-107); 1 = keras Adam, 2 / 3 = keras SGD with / without Nesterov momentum -- the two other optimizers the
// reference trainer can select (yolov3/trainer.py:70-73: SGD(momentum=0.95, nesterov=True), Adam(amsgrad=True)).  Their update rules are
// tf.keras' (tensorflow 1.13.1, python/keras/optimizers.py SGD.get_updates / Adam.get_updates; not part of /root/reference):
//   SGD:  lr' = lr / (1 + decay * iterations);  v = momentum * m - lr' * g;  m <- v;  p += nesterov ? momentum * v - lr' * g : v
//   Adam: t = iterations + 1;  lr_t = lr' * sqrt(1 - b2^t) / (1 - b1^t);  m, v as RAdam;  p -= lr_t * m / (sqrt(amsgrad ? max(vhat, v) : v) + eps)
__global__ void radam_schedule_kernel(float* __restrict__ f, long long* __restrict__ it, int kind, float beta1, float beta2, float decay,
                                      float warmup_coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (kind != 0) {
    double lr = (double)f[0];
    if (decay > 0.f) lr = lr * (1.0 / (1.0 + (double)decay * (double)it[0]));
    it[0] += 1;
    const double t = (double)it[0];
    f[1] = kind == 1 ? (float)(lr * (sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)))) : (float)lr;
    f[2] = 0.f;
    f[3] = (float)kind;
    return;
  }
  // The reference evaluates this scalar chain in float32 (K.floatx()); there 1 - beta_2^t cancels catastrophically for small t
  // (rho_t = rho_inf - 2t*b2^t/(1-b2^t) is a difference of two ~2e3 numbers), so float32 results differ by up to ~1 % between
  // pow() implementations.  We evaluate in double -- the value every float32 implementation approximates -- and round once.
  double lr = (double)f[0];
  if (decay > 0.f) lr = lr * (1.0 / (1.0 + (double)decay * (double)it[0]));  // radam.py:61-64 (pre-increment counter)
  it[0] += 1;                                                                  // radam.py:66
  const double t = (double)it[0];
  const double b1 = (double)beta1, b2 = (double)beta2;
  const double b1p = pow(b1, t), b2p = pow(b2, t);                             // radam.py:77-78
  const double rho_inf = 2.0 / (1.0 - b2) - 1.0;                               // radam.py:54
  const double rho_t = rho_inf - 2.0 * t * b2p / (1.0 - b2p);                  // radam.py:79
  double lr_t;
  if (rho_t >= 5.0)                                                            // radam.py:81-85
    lr_t = sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t)) * lr * (sqrt(1.0 - b2p) / (1.0 - b1p));
  else
    lr_t = (double)warmup_coef * lr / (1.0 - b1p);
  f[1] = (float)lr_t;
  f[2] = (float)rho_t;
  f[3] = rho_t >= 5.0 ? 1.f : 0.f;
}

constexpr int OPT_THREADS = 256;
constexpr int OPT_CHUNK = 256;  // elements per l2-table entry; every variable's slot is padded to a multiple of this

__global__ __launch_bounds__(OPT_THREADS) void radam_l2_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                               float* __restrict__ v, float* __restrict__ vhat, bf16_t* __restrict__ pb,
                                                               const float* __restrict__ l2_table, size_t n4, const float* __restrict__ sched,
                                                               float beta1, float beta2, float eps, float grad_scale, int zero_grad,
                                                               float* __restrict__ l2_partial, int* __restrict__ nonfinite) {
  const float lr_t = sched[1];
  const int rule = (int)sched[3];
  const bool adaptive = rule == 1;
  float l2acc = 0.f;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * OPT_THREADS) {
    const float lam = l2_table[(i * 4) / OPT_CHUNK];
    float4 P = reinterpret_cast<float4*>(p)[i];
    float4 G = reinterpret_cast<float4*>(g)[i];
    float4 Mv = reinterpret_cast<float4*>(m)[i];
    float4 V = reinterpret_cast<float4*>(v)[i];
    float4 VH = vhat ? reinterpret_cast<float4*>(vhat)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {Mv.x, Mv.y, Mv.z, Mv.w}, vv[4] = {V.x, V.y, V.z, V.w};
    float hh[4] = {VH.x, VH.y, VH.z, VH.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      l2acc += lam * pp[j] * pp[j];
      if (nonfinite && !__builtin_isfinite(gg[j])) { gg[j] = 0.f; bad = true; }   // an overflowed (fp16) / NaN gradient element: not applied, counted
      const float gr = gg[j] * grad_scale + 2.f * lam * pp[j];
      if (rule >= 2) {                                               // keras SGD: beta1 is the momentum, m the velocity
        const float vel = beta1 * mm[j] - lr_t * gr;
        mm[j] = vel;
        pp[j] = rule == 2 ? pp[j] + beta1 * vel - lr_t * gr : pp[j] + vel;
        continue;
      }
      mm[j] = beta1 * mm[j] + (1.f - beta1) * gr;                    // radam.py:88
      vv[j] = beta2 * vv[j] + (1.f - beta2) * (gr * gr);             // radam.py:89
      float den = vv[j];
      if (vhat) { hh[j] = fmaxf(hh[j], vv[j]); den = hh[j]; }         // radam.py:91-94
      pp[j] = pp[j] - lr_t * (adaptive ? mm[j] / (sqrtf(den) + eps) : mm[j]);   // radam.py:93/96
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (vhat) reinterpret_cast<float4*>(vhat)[i] = make_float4(hh[0], hh[1], hh[2], hh[3]);
    if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pb) {
      uint2 o;
      o.x = pack_bf2(pp[0], pp[1]);
      o.y = pack_bf2(pp[2], pp[3]);
      reinterpret_cast<uint2*>(pb)[i] = o;
    }
  }
  if (nonfinite && __any(bad) && (threadIdx.x & 63) == 0) atomicAdd(nonfinite, 1);
  __shared__ float red[OPT_THREADS / 64];
  const float s = wave_sum(l2acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && l2_partial) l2_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(OPT_THREADS) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * OPT_THREADS) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    uint2 o;
    o.x = pack_bf2(a.x, a.y);
    o.y = pack_bf2(a.z, a.w);
    reinterpret_cast<uint2*>(y)[i] = o;
  }
}

// out[0] = sum(partial[0..n)) (+ add[0] if add); out_plain[0] = the sum alone (if given)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n, const float* __restrict__ add,
                                                           float* __restrict__ out, float* __restrict__ out_plain) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    if (out_plain) out_plain[0] = t;
    out[0] = t + (add ? add[0] : 0.f);
  }
}

}  // namespace
int g_opt_wgs = 2048;      // "opt_wgs" tuning: most workgroups of the optimizer / cast launches (grid-stride loops)
namespace {

inline int opt_grid(size_t n4) {
  size_t b = (n4 + OPT_THREADS - 1) / OPT_THREADS;
  if (b > (size_t)g_opt_wgs) b = (size_t)g_opt_wgs;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int yolo_radam_schedule(float* sched, int64_t* iterations, float beta1, float beta2, float decay, float warmup_coef, void* stream) {
  YOLO_CHECK_ARG(sched && iterations, "null pointer");
  hipLaunchKernelGGL(radam_schedule_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sched, (long long*)iterations, 0, beta1, beta2, decay,
                     warmup_coef);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_optimizer_schedule(float* sched, int64_t* iterations, int kind, float beta1, float beta2, float decay, void* stream) {
  YOLO_CHECK_ARG(sched && iterations, "null pointer");
  YOLO_CHECK_ARG(kind >= 0 && kind <= 3, "kind: 0 RAdam, 1 Adam, 2 SGD momentum + Nesterov, 3 SGD momentum");
  hipLaunchKernelGGL(radam_schedule_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sched, (long long*)iterations, kind, beta1, beta2, decay,
                     1.f);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_radam_l2_blocks(int64_t n) { return n > 0 && n % OPT_CHUNK == 0 ? opt_grid((size_t)n / 4) : YOLO_ERR_INVALID_ARG; }

extern "C" int yolo_radam_l2_step(float* params, float* grads, float* m, float* v, float* vhat, void* params_bf16, const float* l2_table,
                                  int64_t n, const float* sched, float beta1, float beta2, float eps, float grad_scale, int zero_grad,
                                  float* l2_partial, int* nonfinite, void* stream) {
  YOLO_CHECK_ARG(params && grads && m && v && l2_table && sched, "null pointer");
  YOLO_CHECK_ARG(n > 0 && n % OPT_CHUNK == 0, "n must be a positive multiple of 256 (slots are padded)");
  const size_t n4 = (size_t)n / 4;
  hipLaunchKernelGGL(radam_l2_kernel, dim3(opt_grid(n4)), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, grads, m, v, vhat,
                     (bf16_t*)params_bf16, l2_table, n4, sched, beta1, beta2, eps, grad_scale, zero_grad, l2_partial, nonfinite);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  YOLO_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "bad argument");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(opt_grid((size_t)n / 4)), dim3(OPT_THREADS), 0, (hipStream_t)stream, x, (bf16_t*)y, (size_t)n / 4);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_sum_partials(const float* partial, int n, const float* add, float* out, float* out_plain, void* stream) {
  YOLO_CHECK_ARG(partial && out && n > 0, "bad argument");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, add, out, out_plain);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
