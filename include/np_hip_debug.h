/*
 * np_hip_debug.h — the probe side of libnp_hip.so: what tests/, tools/ and bench.py use to FORCE a kernel variant, to look
 * at a planner decision, to exercise the rendezvous / transport without peers, or to inject an error.  A binding of the
 * back end includes np_hip.h only and never calls anything declared here.
 *
 * PROCESS-GLOBAL STATE.  Every np_*_set_variant below flips a process-wide switch that stays flipped until it is set back
 * (0 / the documented "default" value): all threads and all devices of the process see it, none of them is thread-safe
 * against a concurrent launch, and a test that sets one restores it in a `finally`.  Results stay inside the documented
 * bounds under every variant — they select among kernels that compute the same thing — but bit patterns of reductions
 * and GEMMs may differ between variants (different summation order).  The np_*_debug_* / np_debug_* functions keep no
 * state except np_comm_debug_loopback (on until switched off) and np_debug_raise_device_error (raises the current
 * device's error word, which stays up until np_clear_device_error()).
 */
#ifndef NUMPOWER_AMD_NP_HIP_DEBUG_H
#define NUMPOWER_AMD_NP_HIP_DEBUG_H

#include "np_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How the communicator orders its two streams and issues the sharded GEMM (tuning / A-B; np_comm.hip):
 *   0  default: device-side flags (a one-lane kernel publishes a sequence number on the producing stream, a one-lane
 *      kernel on the consuming stream waits for it).  On a one-rank communicator the slab is ONE progress-reporting
 *      GEMM launch (piece c's transfer is released by the GEMM's own tile counter); with peers it is one GEMM launch
 *      per piece, a kernel boundary behind every piece — the single launch has only ever run on one GPU, so with
 *      world > 1 it is opt-in (3).  Falls back to HIP events by itself if the self-test at np_comm_init finds that
 *      flags do not get through (both streams on one hardware queue).
 *   1  HIP events (hipEventRecord + hipStreamWaitEvent) and one GEMM launch per piece
 *   2  device-side flags, one GEMM launch per piece
 *   3  device-side flags, ONE progress-reporting GEMM launch per slab at any world size
 * np_comm_sync_mode: 1 = flags in use, 0 = events, -1 = no communicator. */
int np_comm_set_variant(int variant);
int np_comm_sync_mode(void);

/* testing: the rendezvous of np_comm_init alone (no device, no RCCL) — rank 0's 128 bytes reach every peer */
int np_comm_debug_exchange(int rank, int world, const char *endpoint, void *bytes128, double timeout_s);
/* testing: the exchange np_sgemm_strided_batched_allgather makes rank `rank` of `world` issue for a slab of `slab` items
 * of item_bytes in `chunks` pieces (point to point), computed by the same functions as the real path, without a device or
 * a communicator: records {piece, send to, send offset, bytes, receive from, receive offset} (byte offsets into the
 * replicated result); *host_count = number of records (also when max_records is smaller). */
int np_comm_debug_plan(int rank, int world, size_t slab, size_t item_bytes, int chunks, unsigned long long *host_out,
                       size_t max_records, size_t *host_count);
/* the step model behind np_sgemm_strided_batched_allgather(chunks = 0) (np_comm.hip: model_pieces), for any
 * world / shape on a device of `cus` CUs (0 = 256), without a device or a communicator: *host_chunks = the piece count it
 * picks; host_ms5 (may be NULL) = the modelled step in ms with 1, 2, 4, 8, 16 pieces. */
int np_comm_debug_model(int world, size_t slab, size_t M, size_t N, size_t K, int cus, int *host_chunks, double *host_ms5);
/* testing: dst <- src through one grouped ncclSend / ncclRecv pair from this rank to itself on the communication
 * stream, then np_comm_wait() — the P2P transport on a box with a single GPU */
int np_comm_debug_sendrecv_self(const void *dev_src, void *dev_dst, size_t bytes);
/* testing: `count` (<= 64) self transfers of `bytes` on the communication stream, not ordered behind the library stream (so
 * they compete with whatever that stream is running), each bracketed by its own events; host_ms[i] = duration of transfer i.
 * Returns when the communication stream has drained. */
int np_comm_debug_loopback_timed(const void *dev_src, void *dev_dst, size_t bytes, int count, float *host_ms);
/* testing: every piece gathered point-to-point is from now on ALSO sent from this rank to itself into dev_scratch
 * (pieces larger than `bytes` are not): real RCCL traffic next to the GEMM on a box without a peer.  (NULL, 0) = off. */
int np_comm_debug_loopback(void *dev_scratch, size_t bytes);

/* Kernel-variant selection for tuning/benchmarks (0 = default heuristic; -1 = whole-K plans only (no split-K, no
 * stream-K), -2 = default planner again, -3 = default planner + always pad unaligned operands, -4 = stream-K wherever
 * the kernel can run it, -5 = default planner without stream-K, -6 = operands whose rows are not float4-loadable never
 * go to the LDS-DMA kernel as they are (padded copies / register-staged kernels, as before round 3), -8 = they always do, whatever the size, -7 = back to the
 * default: from a size threshold up), -9 = never peel a thin ragged edge (M % 256 <= 32 rows, N % 128 <= 2 columns) off a
 * large product, -11 = always when there is one, -10 = back to the default: when the planner's model says it pays), -12 = products with M <= 64 rows go to the tiled
 * kernels instead of sgemm_fewrows_kernel / sgemm_skinny_kernel (as before round 3), -13 = back; round 4: -14 / -15 = plans
 * without / with the mid-size LDS-DMA tiles (sgemm_dmas_kernel), -16 / -17 = their tiles walked row-major / in XCD-aware bands,
 * -18 / -19 = ragged whole-K 64 x 64 products on four / eight waves, -20 / -21 = plans without / with the k-quartered tiles
 * (sgemm_kq_kernel); -(1000 + 100 * shape + S) forces sgemm_dmas_kernel's tile `shape` with K split S ways wherever it applies,
 * -(2000 + shape) forces sgemm_kq_kernel's tile `shape` (0 .. 6: 48x48, 32x32, 64x64, 48x32, 64x32, 64x48, 80x48), -999 ends
 * either forcing; round 5, deep-K products of a few tiles: -22 / -23 = plans without / with K-chunks on the k-quartered tiles,
 * -24 = that plan wherever one exists, -25 / -26 = its launch tile-major / chunk-major over the XCDs, -(30000 + 1000 * shape + S)
 * forces single products of K >= 512 onto S K-chunks of k-quartered tile `shape` (-30000: off), -(40 + q), q = 0 .. 59: a thin
 * K-chunk launch (17 .. 2047 rows, N <= 64, K >= 16384) of fewer than q / 4 workgroups per CU goes to the planner instead
 * (0 = never, as before round 5; default 59).  All of them are process-wide setters for A/B measurements and tests: results stay within the same bounds. */
int np_runtime_set_variant(int variant);   /* how host-result calls wait: 0 = hipStreamSynchronize, 1 = spin on a stream-written flag, 2 = spin on the result itself (default) */
int np_sgemm_set_variant(int variant);
/* debug: the planner's choice for one dense, aligned M x N x K product on a device of `cus` CUs (0 = the current device; any
 * other value needs no device: the planner is host arithmetic).  out[11]: cfg, tail_rows, S, modelled us of the tiled plan;
 * stream-K taken (0/1), its modelled us; cfg, S, us of the best mid-size-tile plan; cfg, us of the best plan without them
 * (1e300 = does not apply).  tests/test_abi_and_host_cpu.py pins the choices that matter (no reference counterpart:
 * linalg.c:75-79 hands every product to cblas / cuBLAS). */
int np_sgemm_debug_plan(size_t M, size_t N, size_t K, size_t batch, int cus, double *out);
int np_elementwise_set_variant(int variant);   /* launch shape of the streaming kernels (np_elementwise.hip cfg_from_variant) and A/B switches of single kernels, e.g. 9000 = np_binary(pow) with its log2 table in LDS instead of registers (same bits) */
int np_layout_set_variant(int variant);   /* transpose tile: 0 = default, 64, 128; 1 = default tiles without the write-aligned form for output rows off the 128-byte grid */
int np_select_set_variant(int variant);   /* order statistics: 0 = plain three passes only (no bracket path, no one-workgroup kernel), 1 = default (n >= 2^26), else the smallest n that takes it */
int np_select_last_path(int *path);       /* tests / tools: 1 if the last selection ran over the bracket's copied keys, 0 if over the array, 2 if in the one-workgroup kernel (synchronises) */
int np_reduce_set_variant(int variant);   /* streaming reductions, first pass: workgroups per CU (0 = default) */
/* tools: the shader clock in MHz a ~20 us probe kernel sees on the library stream right now (synchronises the stream) */
int np_debug_clock_mhz(float *host_mhz);
/* tools: host_out2[2 w] = HW_REG_HW_ID, host_out2[2 w + 1] = HW_REG_XCC_ID of workgroup w of a `workgroups`-wide launch of one-wave
 * workgroups on the library stream (each spins ~10 us first): where a (CU-masked) stream's work lands (tools/cu_mask_probe.py) */
int np_debug_hw_ids(unsigned *host_out2, size_t workgroups);
/* tools: the CU count every planner (GEMM tiles / stream-K, streaming grids) assumes from now on; 0 = the device's own.  For a CU-masked
 * library stream (np_set_stream): a product planned for 256 CUs runs a partial extra round of workgroups on 248.  Process-global. */
int np_debug_set_cus(int cus);
/* testing: a one-lane kernel on the library stream raises `bits` in the current device's error word — what a device-side wait
 * that gives up does (1 = a stream-ordering wait of np_comm, 2 = a stream-K finisher).  np_sync / np_memcpy_d2h / host-result
 * calls / np_comm_* calls on this device return NP_ERR_DEVICE from then on, until np_clear_device_error(). */
int np_debug_raise_device_error(unsigned bits);
/* testing: kernel launches the library has issued since it was loaded (all devices, all threads): the difference around an
 * expression is how many launches it cost — `nd::exp($a) * $b + 2` through the pending chains of INTEGRATION.md 2c: one. */
int np_debug_launch_count(unsigned long long *host_count);

#ifdef __cplusplus
}
#endif

#endif /* NUMPOWER_AMD_NP_HIP_DEBUG_H */
