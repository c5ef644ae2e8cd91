/* yl_amd_diag.h -- diagnostic entry points of libyl_amd.so that are NOT part of the product ABI (include/yl_amd.h).
 *
 * They exist for the measurement scripts under tools/ (tools/README.md) and for two runtime self-checks; nothing on the
 * predict path calls them, and a binder of the reference's API does not need them.  Same conventions as yl_amd.h: 0 on
 * success, negative on failure, yl_last_error() for the message; `stream` is a hipStream_t passed as void*.  No reference
 * counterpart (the reference runs everything on cuda:0's default stream, engine/predictor.py:149).
 */
#ifndef YL_AMD_DIAG_H
#define YL_AMD_DIAG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Runtime self-check: replays a captured graph of two parallel branches, each one wave spinning for spin_us
 * microseconds.  *ms_graph = best of three replays; *concurrent = 1 when the branches overlapped (time < 1.5 x
 * spin_us), 0 when the HIP runtime serialised them.  Diagnostic only: the hot path's graph is captured as ONE chain
 * (side lanes are opt-in, YL_SIDE_LANES=1) and overlaps batches through several streams, one hardware queue each
 * (GPU_MAX_HW_QUEUES, see _lib.RUNTIME_KNOBS). */
int yl_graph_branch_probe(void* stream, int spin_us, float* ms_graph, int* concurrent);
/* A HIP stream confined to the compute units of `cu_mask` (hipExtStreamCreateWithCUMask).  Measured on MI355X
 * (`yl_xcc_census`, tools/xcd_probe.py): word c of the 8-word mask selects CU c of every (shader engine, XCD) pair; the
 * dispatcher still spreads a launch over all eight XCDs, so a stream can be given a SLICE of every XCD, not an XCD.  No reference counterpart (the reference runs on cuda:0's default
 * stream, engine/predictor.py:149).  Destroy with yl_stream_destroy after synchronising it. */
int yl_stream_create_masked(const uint32_t* cu_mask, int n_words, void** stream_out);
int yl_stream_destroy(void* stream);
/* Launches n_workgroups one-wave workgroups on `stream`; each adds 1 to hist_device[xcd * 64 + shader engine * 16 + cu]
 * of the unit it ran on (512 x uint32 on the device, zeroed by the caller). */
int yl_xcc_census(unsigned* hist_device, int n_workgroups, void* stream);
/* In-kernel phase stamps (development): device_buffer receives {shader clock, 100 MHz clock} pairs from the instrumented
 * kernels of a library built with -DYL_STAMPS ([workgroup][wave][32 phases][2] x uint64); NULL switches them off.  Returns 1
 * when the library is such a build, 0 when it is the production build (the call then has no effect).  tools/res_stamps.py. */
int yl_diag_set_stamps(void* device_buffer);
#ifdef __cplusplus
}
#endif
#endif /* YL_AMD_DIAG_H */
