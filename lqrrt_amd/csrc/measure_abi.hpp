// Host side of measure.hpp (measurement builds only, -DSTEER_TIMING / -DDARE_TIMING): the device timestamps and sums of k_steer, read back.
// Fragment of engine.hip; not part of include/lqrrt_hip.h.
#ifdef STEER_TIMING
// debug build only (tools/ablate_steer.py): phase timestamps of block 0 of the last steer launch, 100 MHz ticks
extern "C" int lqrrt_debug_steer_ts(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lq::g_steer_ts), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    return 0;
}
extern "C" int lqrrt_debug_loop_hist(unsigned long long* out32) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(lq::g_loop_hist), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    return 0;
}
extern "C" int lqrrt_debug_pro_acc(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(lq::g_pro_acc), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    return 0;
}
extern "C" int lqrrt_debug_blk_acc(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lq::g_blk_acc), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    return 0;
}
extern "C" int lqrrt_debug_step_acc(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lq::g_step_acc), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    return 0;
}
#endif
#ifdef DARE_TIMING
// measurement build only (tools/dare_phases.py): phase sums of dare_lqr, workgroup 0, 100 MHz ticks
extern "C" int lqrrt_debug_dare_acc(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(lq::g_dare_acc), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    return 0;
}
#endif
