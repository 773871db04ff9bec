// queuekernel_lean.hip -- the queue kernel (variant 3) once more, for scenes WITHOUT triangles (NoAccel: BASELINE config 2).
//
// Such a scene never walks a tree: the kernel's time is the brute-force lists inside the batches, one wave's work is a serial
// chain (scalar load -> a dozen dependent vector instructions per primitive), and the launch time falls linearly with the waves
// per SIMD (measured: 134.9 / 64.6 / 45.8 ms at 1 / 2 / 3 workgroups per CU on random_spheres 1280x720x64).  This instantiation
// therefore trades path slots and stack entries nobody needs there for a fourth workgroup per CU: 384 slots x 8 entries
// (LDS < 40 KB) and a register target of 128 (the spills land in the Disney program, which an analytic scene's lambertian /
// metal / glass materials never run).  random_spheres: 45.9 -> 39.3 ms, same image bits.  moptix_api.hip picks it when the
// scene has no triangles.
#define PT_WAVES_PER_SIMD 4
#define PT_KP 96
#define PT_STACKN 8
#define PT_QK_EXPORT(name) name##_lean
#include "queuekernel.hip"
