/* maxicalib.h -- MEASUREMENT code, not product: pure store / load streams that bench.py and tools/*_ceiling.py time to quote a kernel
 * against the bandwidth this GPU actually reaches (profiles/r03_write_ceiling.md, r03_read_ceiling.md).  libmaxicalib.so is built
 * beside libmaxigpu.so (maximilian_amd/csrc/Makefile) and links against it for the runtime plumbing (initialisation, streams, error
 * text: mxg_last_error of maxigpu.h).  Nothing in the product library or the drop-in headers calls into it. */
#ifndef MAXICALIB_H
#define MAXICALIB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Streaming fill of `bytes` at d_dst (8 B/lane or 16 B/lane stores): the measured HBM write
 * ceiling that bench.py reports next to the nominal 8 TB/s. */
int mxg_calib_fill(void *d_dst, size_t bytes, int width, void *stream);
/* The same measurement over the store shapes the bank kernels could use (csrc/calib.hip): a region of `rows` rows of
 * `row_bytes` bytes; pattern 0 = grid-stride fill (`blocks` workgroups, 0 = 2048), 1 = column walk (a lane owns `width`
 * bytes of a row and stores them row after row: the shape of out[n*V + v]), 2 = column walk in two time halves;
 * flavour 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0; block = threads per workgroup; xcd = 1 renumbers the
 * workgroups so that every XCD owns one contiguous eighth of a row.  tools/write_ceiling.py, profiles/r03_write_ceiling.md. */
int mxg_calib_fill_ex(void *d_dst, size_t rows, size_t row_bytes, int width, int flavour, int pattern, int block,
                      int blocks, int xcd, void *stream);
/* The READ side: a pure load stream over `bytes` at d_src.  pattern 0 = grid-stride read; 1 = the fused FFT + MFCC kernel's input
 * stream (persistent wavefronts, groups of 8 consecutive 4096-byte frames per wavefront, one frame ahead in flight) with everything
 * but the loads removed (bytes a multiple of 32 768); width 8 or 16 bytes per lane; flavour 0 plain, 1 non-temporal; `blocks`
 * workgroups of `block` threads; d_sink: 8 bytes that are never written in practice.  tools/read_ceiling.py,
 * profiles/r03_read_ceiling.md. */
int mxg_calib_read_ex(const void *d_src, size_t bytes, int width, int flavour, int pattern, int block, int blocks, void *d_sink,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif
