// Fused Adam step for parameters stacked over instances (one network per agent–env instance).
//
// torch.optim.Adam's update (torch/optim/adam.py, _single_tensor_adam, non-capturable form)
//     exp_avg.lerp_(grad, 1 - beta1)
//     exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//     denom = exp_avg_sq.sqrt() / sqrt(1 - beta2 ** step) + eps
//     param.addcdiv_(exp_avg, denom, value = -(lr / (1 - beta1 ** step)))
// runs as ~10 elementwise kernels per parameter tensor, each streaming the stacked tensors through
// HBM again (8 192 instances x 4 932 float64 parameters = 323 MB per tensor role).  Here one
// kernel reads p, g, m, v once and writes p, m, v once — and, if asked, blends the new parameters
// into the target network's copy on the way (w_t += tau (w - w_t)), which the reference does on
// the host weight by weight after every step (agent/dqn.py:366-371).  Every instance has its own step count
// (steps[inst], already incremented for this step) and may be switched off by `active`
// (instances that have finished their trials keep parameters AND optimizer state untouched, as if
// their own single-instance run had simply not executed this step).
//
// Replaces, on the DQN path, what the reference does through
// TorchNetwork.train_on_batch -> optimizer.step() (network/network_torch.py:160-167).
#include "cobel_common.h"

namespace {

// grid = (instances, chunks of the instance's elements): a workgroup works inside one instance, so
// the bias corrections (two pow() and a sqrt in float64) are formed once per thread, not per element.
template <typename T>
__global__ __launch_bounds__(256) void k_adam(T* __restrict__ p, const T* __restrict__ g,
                                              T* __restrict__ m, T* __restrict__ v,
                                              const double* __restrict__ steps,
                                              const uint8_t* __restrict__ active, int64_t per_inst,
                                              double lr, double b1, double b2, double eps, double wd,
                                              T* __restrict__ target, double tau) {
  const int64_t inst = blockIdx.x;
  if (active && !active[inst]) return;
  const double t = steps[inst];
  const T bc1 = (T)(1.0 - pow(b1, t));
  const T bc2_sqrt = (T)sqrt(1.0 - pow(b2, t));
  const T step_size = (T)lr / bc1;
  const int64_t base = inst * per_inst;
  const int64_t stride = (int64_t)gridDim.y * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; k < per_inst; k += stride) {
    const int64_t e = base + k;
    const T pe = p[e];
    T ge = g[e];
    if (wd != 0.0) ge = ge + (T)wd * pe;
    const T mo = m[e], vo = v[e];
    const T mn = mo + (T)(1.0 - b1) * (ge - mo);
    const T vn = vo * (T)b2 + ((T)(1.0 - b2) * ge) * ge;
    const T denom = sqrt(vn) / bc2_sqrt + (T)eps;
    const T pn = pe - step_size * (mn / denom);
    m[e] = mn;
    v[e] = vn;
    p[e] = pn;
    if (target) {   // w_target += tau * (w_online - w_target)  (agent/dqn.py:366-371)
      const T te = target[e];
      target[e] = te + (T)tau * (pn - te);
    }
  }
}

}  // namespace

extern "C" int cobel_adam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq,
                               const double* steps, const uint8_t* active, int64_t n_instances,
                               int64_t per_instance, int32_t is_float64, double lr, double beta1,
                               double beta2, double eps, double weight_decay, void* target,
                               double tau, void* stream) {
  COBEL_REQUIRE(param && grad && exp_avg && exp_avg_sq && steps, COBEL_E_ARG,
                "cobel_adam_step: NULL tensor");
  COBEL_REQUIRE(n_instances >= 0 && per_instance > 0, COBEL_E_RANGE,
                "cobel_adam_step: bad sizes %lld x %lld", (long long)n_instances,
                (long long)per_instance);
  if (n_instances == 0) return COBEL_OK;
  COBEL_REQUIRE(n_instances <= 0x7fffffffLL, COBEL_E_RANGE, "cobel_adam_step: %lld instances",
                (long long)n_instances);
  const int64_t chunks = (per_instance + 1023) / 1024;    // <= 4 elements per thread
  const dim3 grid((unsigned)n_instances, (unsigned)(chunks < 65535 ? chunks : 65535));
  hipStream_t st = (hipStream_t)stream;
  if (is_float64)
    hipLaunchKernelGGL(k_adam<double>, grid, dim3(256), 0, st, (double*)param, (const double*)grad,
                       (double*)exp_avg, (double*)exp_avg_sq, steps, active, per_instance, lr,
                       beta1, beta2, eps, weight_decay, (double*)target, tau);
  else
    hipLaunchKernelGGL(k_adam<float>, grid, dim3(256), 0, st, (float*)param, (const float*)grad,
                       (float*)exp_avg, (float*)exp_avg_sq, steps, active, per_instance, lr, beta1,
                       beta2, eps, weight_decay, (float*)target, tau);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
