/* fo1_ab.h — A/B, ablation and determinism-pin switches of libfo1hip_ab.so.  NOT part of the product ABI.
 *
 * Every function here writes PROCESS-GLOBAL state: not thread-safe, never to be called by a serving process.  They exist so that
 *   - the parity tests can pin the GEMM tile / split-K / GEMV routing (a row's fp32 summation order then does not depend on how many
 *     rows share the launch: "packed pass == one-image passes, bit for bit");
 *   - the measured-slower kernel forms kept for A/B (four-phase and persistent 256 x 256 GEMM schedules, the v_dot2 decode GEMV, the
 *     one-workgroup-per-head decode attention, the round-1 worst-case-grid HFRE gather) can still be selected and re-measured.
 * They are compiled ONLY into the test / bench build (hipcc -DFO1_ENABLE_AB -> vlm_fo1_amd/libfo1hip_ab.so, loaded when FO1_AB=1; the
 * GPU test session sets it in tests/conftest.py).  The product library vlm_fo1_amd/libfo1hip.so exports none of these symbols, carries
 * none of that mutable state (the defaults are compile-time constants) and none of those kernels: tests/test_abi.py checks both
 * export tables against the two headers. */
#ifndef FO1_AB_H
#define FO1_AB_H

#include "fo1.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default) /* libfo1hip*.so are built with -fvisibility=hidden: exactly the declarations of this header are exported */
#endif

/* ---- HFRE gather ---- */
/* footprint pixels one workgroup streams per row-slice (0 = auto: 256 up to 48 boxes, else 512). */
int fo1_hfre_set_pixel_budget(int pixels);
/* fo1_hfre_region_pool_ex: unroll = 8 | 16 independent 16-byte loads per lane (| 32 = scalar finish kernel); chunk = channels per workgroup
 * (power of two, 64..512); budget = pixels per row slice (0 = keep); grid = workgroups walking the work list (0 = keep).
 * Results are bit-identical across unroll / grid (the fp32 sum order depends on budget and chunk only). */
int fo1_hfre_set_tuning(int unroll, int chunk, int budget, int grid);

/* ---- GEMM ---- */
/* staging 0 auto / 1 register-staged / 2 LDS-DMA two-stage / 3, 4, 6 LDS-DMA ring of that depth (counted vmcnt; 6 only for the 64x64 tile,
 * else 4); tile 0 auto / 1 128x128 / 2 64x128 / 3 64x64 / 4 128x256 (8 waves) / 5 256x256; split-K 0 auto / n forced.
 * (2, 1) + splitk 1 + gemv 0 is the determinism pin of the parity tests. */
int fo1_gemm_set_variant(int staging, int tile);
int fo1_gemm_set_splitk(int splits);
int fo1_gemm_set_gemv(int on);   /* M <= 4 goes to the weight-streaming GEMV kernel (default on) */
/* 256x256 kernels: tile rows per group of the XCD-grouped tile order (an XCD's ~32 concurrent tiles = rows x 32 / rows tile columns);
 * 0 = the product rule (2; 8 until round 5).  Bit-identical results for every value: a permutation of the output tiles. */
int fo1_gemm_set_group_m(int rows);
/* 256x256 kernel, bit field: bit 0 = two fat phases per K tile with the DMA issued between MFMAs (0 = four phases); bit 1 = fragment-shaped
 * epilogue stores (0 = LDS-staged coalesced); bit 2 = persistent tile loop (bit-identical, measured 2-5 % slower); bit 3 = non-temporal
 * epilogue stores (no effect measured).  Default 1. */
int fo1_gemm_set_big_schedule(int sched);
/* ablation, RESULTS INVALID: 1 no global loads, 2 no MFMA, 4 no LDS reads + MFMA; 256x256 two-phase kernel: 8 epilogue computed but not
 * stored, 16 one K tile per output tile (profiles/r02_gemm_t0_study.md).
 * Bits 6..12 (round 4, scripts/gemm_loop_ablation.py; plain / residual bf16 products of the 256x256 two-phase kernel only): ABL = bits >> 6 —
 * 1 no LDS-DMA issue inside the K loop, 2 no fragment reads inside it, 4 no barriers, 8 no vmcnt waits, 16 every DMA piece from K tile 0,
 * 64 only the A half of the pieces issued (all RESULTS INVALID), 32 half of a wave's pieces issued in its load-Y segment (results valid);
 * instantiated: 1, 2, 4, 8, 16, 24, 32, 64. */
int fo1_gemm_set_debug(int bits);
/* bit 5 (32) of fo1_gemm_set_debug, results VALID: waves 0 and 7 of every 256x256-kernel workgroup write s_memrealtime (100 MHz) at kernel entry,
 * first MFMA, end of the K loop, end of the epilogue (slot 6: conversions staged in LDS) and their HW_ID / XCC_ID (upper halves of those two
 * words: s_memtime, low 32 bits, at the first MFMA and at the end of the K loop — the shader cycles and so the clock of the loop) to this device buffer,
 * [workgroups][2][8] uint64
 * (scripts/gemm_timeline.py) */
int fo1_gemm_set_stamp_buffer(void* device_buffer);

/* ---- DaViT ---- */
/* fo1_dwconv3x3_ln_bf16 / _var: 1 (default) = product rule: the sliding-window form (a wave walks 8 pixels of a row — 4 at C = 1024 — with the
 * 3 x 3 window in registers) for C = 128 / 256 / 512 / 1024 when the call has >= 4096 waves of runs, else one wave per pixel; 0 = per-pixel
 * everywhere; 2 = the run form at every size it exists for.  Bit-identical. */
int fo1_dwconv_ln_set_form(int run_form);

/* fo1_channel_attention_bf16 / _var: 1 (default) = Gram matrices and the attention product on the matrix cores (v_mfma_f32_32x32x16_bf16; exact
 * bf16 products, fp32 sums in the MFMA's order), 0 = the fp32 FMA kernels of rounds 1-5 (sequential sums).  Equal up to the order of fp32 additions. */
int fo1_channel_attention_set_impl(int mfma);

/* ---- decode step ---- */
/* fo1_gemv_batch_bf16, v_dot2 kernel: 0 (default) = a lane streams 1 / 2 / 4 weight rows per chunk position by M; 1 = always one row. */
int fo1_gemv_batch_set_rows_per_lane(int rpl);
/* 1 (default) = MFMA skinny GEMM (csrc/decode_mfma.hip: the sequences ride as the 16 columns of v_mfma_f32_16x16x32_bf16); 0 = the v_dot2
 * streaming kernel (M <= 8); 3 = MFMA without any 8-row units; 5 = MFMA with the M <= 8 (HALF) units only (no R8 units for 9..32 sequences).
 * All keep a sequence's numbers independent of the batch it decodes in; MFMA and v_dot2 differ from each other in fp32 summation order. */
int fo1_gemv_batch_set_impl(int impl);
/* 0 (default) = 64-key split-KV partials + combine kernel; 1 = one workgroup per (KV head, sequence), partials merged in LDS (measured
 * slower on MI355X: one CU cannot pull a head's K/V^T fast enough); 2 (round 4, batched decode only) = one workgroup per (KV head, sequence) whose four
 * waves each walk their own 64-key tiles through wave-private LDS images, merged once in LDS (21.8 vs 23.4 us per layer at 128 sequences: the launch is
 * within a third of HBM bandwidth either way). */
int fo1_attention_decode_set_impl(int impl);
/* Keys per chunk of the batched decode attention for more than 32 sequences (decode pool): a multiple of 64 up to 4096, default 1024 = a pool
 * slot's whole context, for which the split kernel writes the output rows itself and no combine launch is made. */
int fo1_attention_decode_set_pool_chunk(int keys);
/* keys per split of the batched decode attention at <= 32 sequences (product: 64 up to ... see decode_batch_chunk in csrc/attention.hip). */
int fo1_attention_decode_set_small_chunk(int keys);

/* ---- measured no-gain kernel forms and instruments (round 5: moved out of the product ABI) ---- */
/* SwiGLU over the split-K planes of fo1_gemm_bf16_partials for the gate/up projection against the 16-row interleaved weight:
 * out[m, f] = bf16(bf16(silu(bf16(g))) * bf16(u)), g / u = sum_z of plane columns 32 (f / 16) + f % 16 and + 16 — fo1_gemm_bf16's act 3 epilogue on
 * planes.  Measured slower than the one-GEMM form in the decode pool (36-49 vs 34 us per layer); llm.DecodePool.SPLITS["gateup"] selects it. */
int fo1_splitk_swiglu_bf16(const float* part, int splits, int M, int N, void* out, int ldo, void* stream);
/* fo1_gemm_bf16 for a weight that few rows (<= 128) stream once per call — the decode pool's gate/up and lm_head: W_tiled is a copy of W [N, K] laid
 * out [N / 128][K / 64][128][64] (made once at load), so a K tile of a column tile is one contiguous 16 KB block.  Same kernel, arithmetic and
 * epilogues as fo1_gemm_bf16 (bit-identical on the same tile shape).  N % 128 == 0, K % 64 == 0. */
int fo1_gemm_bf16_wtiled(const void* A, int lda, const void* W_tiled, const void* bias, const void* residual, int ldr, void* C, int ldc, int M, int N,
                         int K, int act, void* stream);
/* Instrumentation: the clock the matrix pipes sustain on this box (csrc/probe.hip).  A register-resident loop of v_mfma_f32_32x32x16_bf16 on
 * every CU (8 waves per workgroup, `iters` x 32 MFMAs per wave, no memory traffic); out = uint64 [workgroups][2] {shader cycles (s_memtime),
 * 100 MHz ticks (s_memrealtime)}.  operands 0 = zeros, 1 = pseudo-random bf16.  cycles / ticks = the DVFS clock; the dense bf16 peak of the
 * roofline (2.5 PFLOP/s) assumes 2.4 GHz. */
int fo1_mfma_clock_probe(int operands, int iters, int workgroups, void* out, void* sink, void* stream);
/* Instrumentation: moves exactly `bytes` in a named access pattern (0 stream store 16 B / lane, 1 the 256 x 256 GEMM's coalesced epilogue store,
 * 2 stream load, 3 LDS-DMA tile load, 4 / 5 stream store 8 / 4 B per lane) so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated on a
 * known byte count per pattern (MI355X_MICROARCH.md: WRITE_SIZE is uncalibrated; scripts/pmc_calibrate.py, profiles/r06_pmc_calibration.json). */
int fo1_traffic_probe(int mode, void* buf, long long bytes, long long ld, int workgroups /* 0 = 2048 */, void* sink, void* stream);
/* Instrumentation (with fo1_profile_enable): per-shape kernel names in the profile rows instead of one row per kernel. */
int fo1_gemm_profile_shapes(int on);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FO1_AB_H */
