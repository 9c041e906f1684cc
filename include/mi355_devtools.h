/*
 * mi355_devtools.h — diagnostics for tools/ (XCD map, page touch, flag preset).  NOT part of the product
 * library: these entry points exist only in libmi355spmv_dev.so (`make devtools` in navierstokes_amd/csrc),
 * which is libmi355spmv.so plus devtools.hip.  the tools (tools/sim_rank.py, xcc_probe.py, cold_probe.py) load it through MI355_SPMV_LIBRARY.
 */
#ifndef MI355_DEVTOOLS_H
#define MI355_DEVTOOLS_H
#include "mi355_spmv.h"
#ifdef __cplusplus
extern "C" {
#endif
/* diagnostic: host_out[b] = XCD (HW_REG_XCC_ID) that workgroup b of a `wgs`-workgroup launch ran on */
int mi_debug_xcc_map(int wgs, int* host_out);
/* diagnostic: one 4-byte read every stride_bytes of each device array the handle's kernels stream (and of up to two caller
 * buffers, e.g. x and y), then a synchronise.  Behind mi_flush_cache() this brings the address translations back without
 * bringing the data back (one line per stride): it separates "cold caches" from "cold TLB" in a cold-start measurement. */
int mi_debug_touch_pages(mi_csr_t A, int stride_bytes, const void* d_extra0, long long bytes0, const void* d_extra1, long long bytes1);
/* timeline of ONE launch of the multi-window ring kernel on the handle's plan (its relabelled twin's, if it has one; depth 4,
 * non-temporal, unmapped instantiation): host_out[4 * i] = {start, end (100 MHz ticks), XCD, blocks of the run (< 0: plain path, 0: no
 * run)} of workgroup i; *wgs_out = the grid.  d_x / d_y: device vectors in the numbering of the matrix that runs (tools/mring_timeline.py). */
int mi_debug_mring_trace(mi_csr_t A, const double* d_x, double* d_y, int max_wgs, long long* host_out, int* wgs_out);
/* placement experiments (tools/placement_lottery.py): move one device array of the handle's ring path (0 coefficients, 1 ring slots,
 * 2 row pointers, 3 ring plan records) to a fresh allocation (how: 0 hipMalloc, 2 / 3 hipExtMallocWithFlags uncached / fine-grained; 1 =
 * hipDeviceMallocContiguous is refused: it ended in a GPU memory fault here); the old one is freed after the new one exists. */
int mi_debug_move_array(mi_csr_t A, int which, int how, unsigned long long* old_ptr, unsigned long long* new_ptr);
/* run-length experiments on one placement (tools/mring_skew_ab.py): re-plan the handle's multi-window ring kernel with runs alternately
 * skew_pct per cent longer / shorter (the longer dispatched to the older workgroup of every CU) INTO the device arrays it already has */
int mi_debug_mring_replan(mi_csr_t A, int skew_pct, int* table_len, int* longest_run);
/* timeline of ONE one-launch push step through the sliced stream (spmv_sstream_fused with its s_memrealtime stamps compiled in, D = 8,
 * temporal; the MI355_PUSH_LOOPBACK arrangement of tools/sim_rank.py): host_out[4 * g + {0, 1, 2, 3}] = logical workgroup g's start, loop
 * begin, loop end, end (100 MHz ticks); halo_out[g] = (reads ghosts) + 2 * (pushes) + 4 * rounds. */
int mi_debug_part_push_trace(mi_part_t P, double* d_x_ext, double* d_y_local, int max_wgs, long long* host_out, int* wgs_out, int* halo_out);
/* timing experiments on the staged one-launch step of a blocked rank: mode bits 1 nobody waits for the exchange, 2 no push, 4 no window
   copy, 8 no wait for the neighbours.  Results are wrong afterwards: tools only. */
int mi_debug_part_ext_mode(mi_part_t P, int mode);
/* one traced launch of that step: host_out[3 g + {0,1,2}] = {start, wait over, end} of workgroup g in s_memrealtime ticks; modes_out[g] = -1 for an
   exchange workgroup, else the unit's mode bits (1 waits, 2 sixteen lanes per block row) */
int mi_debug_part_ext_trace(mi_part_t P, double* d_x_ext, double* d_y_local, int max_wgs, long long* host_out, int* wgs_out, int* modes_out);
/* development aid (tools/sim_rank.py): preset every flag slot of this rank's window */
int mi_part_push_debug_preset(mi_part_t P, unsigned value);

#ifdef __cplusplus
}
#endif
#endif /* MI355_DEVTOOLS_H */
