// ranenv_step.hip -- the builds of the step kernel for ONE row width NP (-DRANENV_NP=8 / 10 / 16: three objects of this file, compiled in
// parallel) and their entry in the launch table (ranenv_internal.h).  The kernel code is ranenv_step_body.hpp.
#include "ranenv_step_body.hpp"

#ifndef RANENV_NP
#error "compile with -DRANENV_NP=8, 10 or 16"
#endif

namespace {

constexpr int NPW = RANENV_NP;

typedef void (*step_kernel_t)(const KP);

hipError_t go(step_kernel_t kernel, dim3 grid, dim3 block, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const KP &kp)
{
    // (the extended launch costs the host several times an ordinary one: only while profiling)
    if (ev0) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, ev0, ev1, 0, kp);
    else hipLaunchKernelGGL(kernel, grid, block, 0, s, kp);
    return hipSuccess;          // (launch errors are collected by the caller: hipGetLastError)
}

// the build of the step kernel that `l` names, or null where that combination does not exist
step_kernel_t pick(const StepLaunch &l)
{
    const int mode = l.mode & 3;
    const bool pe = (l.mode & MODE_PE) != 0;
    switch (l.build) {
    case SB_LEAN:
        if (pe) {
            if (mode == MODE_STEP) return l.many ? ranenv_core_kernel<MODE_STEP | MODE_PE, NPW, true> : ranenv_core_kernel<MODE_STEP | MODE_PE, NPW, false>;
            if (mode == MODE_DENSE) return ranenv_core_kernel<MODE_DENSE | MODE_PE, NPW, false>;
            return nullptr;
        }
        if (mode == MODE_STEP) return l.many ? ranenv_core_kernel<MODE_STEP, NPW, true> : ranenv_core_kernel<MODE_STEP, NPW, false>;
        if (mode == MODE_DENSE) return ranenv_core_kernel<MODE_DENSE, NPW, false>;
        return ranenv_core_kernel<MODE_RESET, NPW, false>;
    case SB_SMALL:
        if (pe) return nullptr;
        if (mode == MODE_STEP) return l.many ? ranenv_core_kernel_small<MODE_STEP, NPW, true> : ranenv_core_kernel_small<MODE_STEP, NPW, false>;
        if (mode == MODE_DENSE) return ranenv_core_kernel_small<MODE_DENSE, NPW, false>;
        return ranenv_core_kernel_small<MODE_RESET, NPW, false>;
    case SB_GATHER:
        if (mode == MODE_DENSE) return nullptr;
        if (pe) {
            if (mode != MODE_STEP) return nullptr;
            return l.many ? ranenv_core_kernel_gather<MODE_STEP | MODE_PE, NPW, true> : ranenv_core_kernel_gather<MODE_STEP | MODE_PE, NPW, false>;
        }
        if (mode == MODE_STEP) return l.many ? ranenv_core_kernel_gather<MODE_STEP, NPW, true> : ranenv_core_kernel_gather<MODE_STEP, NPW, false>;
        return ranenv_core_kernel_gather<MODE_RESET, NPW, false>;
    case SB_TINY1:
        return (mode == MODE_STEP && !pe && !l.many) ? ranenv_core_kernel_tiny1<NPW> : nullptr;
    case SB_MIXED:
        if (mode != MODE_STEP || pe) return nullptr;
        if (l.gather) return l.many ? ranenv_core_kernel_mixed<NPW, true, true> : ranenv_core_kernel_mixed<NPW, false, true>;
        return l.many ? ranenv_core_kernel_mixed<NPW, true, false> : ranenv_core_kernel_mixed<NPW, false, false>;
    case SB_PACKED:
#if RANENV_NP == 8
        if (mode != MODE_STEP || pe) return nullptr;
        if (l.gather) return l.many ? ranenv_core_kernel_packed<8, true, true> : ranenv_core_kernel_packed<8, false, true>;
        return l.many ? ranenv_core_kernel_packed<8, true, false> : ranenv_core_kernel_packed<8, false, false>;
#else
        return nullptr;
#endif
    case SB_PERSIST:
        return l.gather ? ranenv_persist_kernel<true, NPW> : ranenv_persist_kernel<false, NPW>;
    case SB_PERSIST_TINY:
        return l.gather ? nullptr : ranenv_persist_kernel_tiny<NPW>;
    default:
        return nullptr;
    }
}

}  // namespace

namespace ranenv_dev {

#define RANENV_CAT_(a, b) a##b
#define RANENV_CAT(a, b) RANENV_CAT_(a, b)

hipError_t RANENV_CAT(launch_step_np, RANENV_NP)(const StepLaunch &l, dim3 grid, dim3 block, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, const KP &kp)
{
    const step_kernel_t k = pick(l);
    if (!k) return hipErrorInvalidDeviceFunction;
    return go(k, grid, block, stream, ev0, ev1, kp);
}

const void *RANENV_CAT(step_kernel_ptr_np, RANENV_NP)(const StepLaunch &l) { return reinterpret_cast<const void *>(pick(l)); }

size_t RANENV_CAT(shared_core_bytes_np, RANENV_NP)() { return sizeof(SharedCore<NPW>); }

}  // namespace ranenv_dev
