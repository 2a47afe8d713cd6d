// In-kernel cycle stamps for the DIAGNOSTIC builds under tools/micro/ (compiled with -DPD_STAMP): s_memtime accumulators around the parts
// of a kernel's main loop.  In the product build (no -DPD_STAMP) every macro below expands to nothing: no stamp executes, no register
// or kernel argument exists for them.  Read the SHARES of a stamped build, not its lengths: the fences around each stamp forbid
// overlaps the product build has (cdna_hip_programming.md section 7, "In-kernel stamps").
#pragma once
#ifdef PD_STAMP
#define PD_T_NOW() ([]() { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); __builtin_amdgcn_sched_barrier(0); return t_; }())
#define PD_T_ADD(acc, a, b) acc += (b) - (a)
#define PD_T_ONLY(...) __VA_ARGS__
#else
#define PD_T_NOW() 0ull
#define PD_T_ADD(acc, a, b) do { } while (0)
#define PD_T_ONLY(...)
#endif
