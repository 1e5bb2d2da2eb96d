// EXPERIMENT (round 3, not built): K1 with BOTH packed signals in one thread -- real parts of the two signals in one 64-bit
// register pair, imaginary parts in another, so that every real operation of a butterfly is one v_pk_*_f32 instruction without
// swizzles or moves; passes 1-2 on 120 threads, pass 3 on 100; one 16-byte LDS element per position; 16-byte audio loads.
// Static count: 469 packed + ~35 scalar arithmetic instructions per thread and frame on HALF the threads (round 2: 309 packed +
// 159 scalar + 242 v_mov on all of them) -- the FFT's issue slots are cut by ~45 %.  Measured on MI355X (tools/feat_ab.sh,
// B = 64 x 60 s): 1.74 / 1.63 ms (nhwc8 / nchw7) at three workgroups per CU (168 VGPRs) against 1.78 / 1.62 ms for the round-2
// kernel; 2.35 ms at four workgroups per CU (128 VGPRs: 82 spilled).  Conclusion: K1 is NOT bound by the FFT's instruction
// issue -- halving it moved nothing -- but by the per-frame chain of barriers, LDS round trips and table loads with four
// frames in flight per CU.  Build it as an A/B library with  bash tools/build_variant.sh k1v4 ../../tools/experiments/<this file> -DK1_OCC=3
