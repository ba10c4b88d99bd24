// calib.hip -- measurement kernels for the HBM *write* ceiling of the box (and, further down, the READ ceiling) (no reference counterpart: they exist so that
// bench.py and profiles/r03_write_ceiling.md can quote the voice-bank render against what a pure store stream of the same
// shape reaches on this GPU, next to the 8 TB/s spec).
//
// The bank kernels (K1 osc_kernel, K2f voice_kernel) store out[n*V + v]: a lane owns a voice (a column) and walks down the
// rows, one 8-byte store per sample.  The calibration family covers that shape and its alternatives:
//   pattern 0  grid-stride fill of the flat region (blocks x block threads, the classic memset shape)
//   pattern 1  column walk: thread = `width` bytes of a row, for (n < rows) store -- K1's shape with the arithmetic removed
//   pattern 2  column walk in two time halves (gridDim.y = 2), K1's time split
// width 8 or 16 bytes per lane; store flavour: plain, nt, sc1 (write-through, drops the L2 line), sc0 sc1, sc1 nt, sc0;
// xcd = 1 renumbers the workgroups so that the eight XCDs (workgroup id mod 8) each own one contiguous eighth of a row
// instead of every eighth 2-KB piece.
#include "mxg_common.h"
#include "../../include/maxicalib.h"

namespace mxg {
namespace {

template <int FLAV>
__device__ __forceinline__ void st8(double *p, double v) {
    if constexpr (FLAV == 0) *p = v;
    else if constexpr (FLAV == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (FLAV == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (FLAV == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (FLAV == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int FLAV>
__device__ __forceinline__ void st16(double2v *p, double2v v) {
    if constexpr (FLAV == 0) *p = v;
    else if constexpr (FLAV == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (FLAV == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (FLAV == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (FLAV == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

template <int WIDTH, int FLAV>
__global__ void calib_flat(char *dst, size_t bytes) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * WIDTH;
    const size_t stride = (size_t)gridDim.x * blockDim.x * WIDTH;
    for (; i + WIDTH <= bytes; i += stride) {
        if constexpr (WIDTH == 8) st8<FLAV>((double *)(dst + i), 1.0);
        else st16<FLAV>((double2v *)(dst + i), (double2v){1.0, 2.0});
    }
}

template <int WIDTH, int FLAV>
__global__ void calib_cols(char *dst, size_t rows, size_t row_bytes, int xcd) {
    const unsigned b = xcd_block(blockIdx.x, gridDim.x, xcd);
    const size_t col = ((size_t)b * blockDim.x + threadIdx.x) * WIDTH;
    if (col + WIDTH > row_bytes) return;
    const size_t plen = (rows + gridDim.y - 1) / gridDim.y;
    const size_t nA = blockIdx.y * plen < rows ? blockIdx.y * plen : rows;
    const size_t nB = nA + plen < rows ? nA + plen : rows;
    char *p = dst + nA * row_bytes + col;
    double x = (double)col;
#pragma unroll 4
    for (size_t n = nA; n < nB; n++) {
        x += 1.0;  // a value that changes per row, like a rendered sample
        if constexpr (WIDTH == 8) st8<FLAV>((double *)p, x);
        else st16<FLAV>((double2v *)p, (double2v){x, -x});
        p += row_bytes;
    }
}

typedef void (*flat_fn)(char *, size_t);
typedef void (*cols_fn)(char *, size_t, size_t, int);
template <int W>
flat_fn pick_flat(int f) {
    switch (f) {
        case 1: return calib_flat<W, 1>;
        case 2: return calib_flat<W, 2>;
        case 3: return calib_flat<W, 3>;
        case 4: return calib_flat<W, 4>;
        case 5: return calib_flat<W, 5>;
        default: return calib_flat<W, 0>;
    }
}
template <int W>
cols_fn pick_cols(int f) {
    switch (f) {
        case 1: return calib_cols<W, 1>;
        case 2: return calib_cols<W, 2>;
        case 3: return calib_cols<W, 3>;
        case 4: return calib_cols<W, 4>;
        case 5: return calib_cols<W, 5>;
        default: return calib_cols<W, 0>;
    }
}

// ---- the READ side (round 3): what a pure load stream reaches, for the fused FFT + MFCC kernel's 4 KB-per-frame input ----------
//   pattern 0  grid-stride read of the flat region
//   pattern 1  the fused kernel's stream: persistent wavefronts, a wavefront takes groups of 8 consecutive 4096-byte frames (group
//              g0 + k * waves), a frame as 4096 / (64 * width) loads of `width` bytes per lane at 64 * width-byte spacing, the next
//              frame's loads issued before this frame's values are consumed -- K67 with everything but the loads removed
// width 8 or 16; flavour 0 plain, 1 non-temporal.  Every value is folded into a per-thread sum that is stored only if it hits an
// impossible value, so the loads cannot be removed.
template <int WIDTH, int FLAV>
__device__ __forceinline__ double ld_sum(const char *p) {
    if constexpr (WIDTH == 8) {
        const double v = FLAV ? __builtin_nontemporal_load((const double *)p) : *(const double *)p;
        return v;
    } else {
        const double2v v = FLAV ? __builtin_nontemporal_load((const double2v *)p) : *(const double2v *)p;
        return v.x + v.y;
    }
}
template <int WIDTH, int FLAV>
__global__ void calib_read_flat(const char *src, size_t bytes, double *sink) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * WIDTH;
    const size_t stride = (size_t)gridDim.x * blockDim.x * WIDTH;
    double acc = 0.0;
#pragma unroll 8
    for (; i + WIDTH <= bytes; i += stride) acc += ld_sum<WIDTH, FLAV>(src + i);
    if (acc == 1.2345e300) sink[0] = acc;
}
template <int WIDTH, int FLAV>
__global__ void calib_read_frames(const char *src, size_t nframes, double *sink) {
    constexpr int L = 4096 / (64 * WIDTH);  // loads per lane and frame
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6), w0 = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t ngroups = nframes / 8;
    double acc = 0.0, cur[L], nxt[L];
    auto load = [&](size_t fr, double (&d)[L]) {
        const char *x = src + (fr < nframes ? fr : nframes - 1) * 4096 + (size_t)lane * WIDTH;
#pragma unroll
        for (int e = 0; e < L; e++) d[e] = ld_sum<WIDTH, FLAV>(x + (size_t)e * 64 * WIDTH);
    };
    load(w0 * 8, nxt);
    for (size_t g = w0; g < ngroups; g += waves) {
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int e = 0; e < L; e++) cur[e] = nxt[e];
            load(j + 1 < 8 ? g * 8 + j + 1 : (g + waves) * 8, nxt);
#pragma unroll
            for (int e = 0; e < L; e++) acc += cur[e];
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

}  // namespace
}  // namespace mxg

extern "C" int mxg_calib_read_ex(const void *d_src, size_t bytes, int width, int flavour, int pattern, int block, int blocks,
                                 void *d_sink, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_src && d_sink && (width == 8 || width == 16), "width is 8 or 16");
    MXG_REQUIRE(flavour >= 0 && flavour <= 1 && pattern >= 0 && pattern <= 1, "unknown flavour / pattern");
    MXG_REQUIRE(block >= 64 && block <= 1024 && (block & 63) == 0 && blocks > 0, "block is a multiple of 64 up to 1024, blocks > 0");
    MXG_REQUIRE((((uintptr_t)d_src) & 15) == 0 && (pattern == 0 || bytes % 32768 == 0), "misaligned region / not whole groups of 8 frames");
    if (bytes == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    KernelTimer kt("calib_read", st);
    const dim3 grid((unsigned)blocks), blk((unsigned)block);
#define MXG_RD(K, ...)                                                                                    \
    do {                                                                                                  \
        if (width == 8) {                                                                                 \
            if (flavour) hipLaunchKernelGGL((K<8, 1>), grid, blk, 0, st, __VA_ARGS__);                     \
            else hipLaunchKernelGGL((K<8, 0>), grid, blk, 0, st, __VA_ARGS__);                             \
        } else {                                                                                          \
            if (flavour) hipLaunchKernelGGL((K<16, 1>), grid, blk, 0, st, __VA_ARGS__);                    \
            else hipLaunchKernelGGL((K<16, 0>), grid, blk, 0, st, __VA_ARGS__);                            \
        }                                                                                                 \
    } while (0)
    if (pattern == 0) MXG_RD(calib_read_flat, (const char *)d_src, bytes, (double *)d_sink);
    else MXG_RD(calib_read_frames, (const char *)d_src, bytes / 4096, (double *)d_sink);
#undef MXG_RD
    return check_hip(hipGetLastError(), "calib_read_ex launch");
}

extern "C" int mxg_calib_fill_ex(void *d_dst, size_t rows, size_t row_bytes, int width, int flavour, int pattern, int block,
                                 int blocks, int xcd, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_dst && (width == 8 || width == 16), "width is 8 or 16");
    MXG_REQUIRE(flavour >= 0 && flavour <= 5 && pattern >= 0 && pattern <= 2, "unknown flavour / pattern");
    MXG_REQUIRE(block >= 64 && block <= 1024 && (block & 63) == 0, "block is a multiple of 64 up to 1024");
    MXG_REQUIRE(row_bytes % (size_t)width == 0 && (((uintptr_t)d_dst) & 15) == 0, "misaligned region");
    if (rows == 0 || row_bytes == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    KernelTimer kt("calib_fill", st);
    if (pattern == 0) {
        flat_fn fn = width == 8 ? pick_flat<8>(flavour) : pick_flat<16>(flavour);
        hipLaunchKernelGGL(fn, dim3((unsigned)(blocks > 0 ? blocks : 2048)), dim3((unsigned)block), 0, st, (char *)d_dst,
                           rows * row_bytes);
    } else {
        cols_fn fn = width == 8 ? pick_cols<8>(flavour) : pick_cols<16>(flavour);
        const size_t lanes = row_bytes / (size_t)width;
        hipLaunchKernelGGL(fn, dim3((unsigned)((lanes + block - 1) / block), pattern == 2 ? 2u : 1u), dim3((unsigned)block), 0, st,
                           (char *)d_dst, rows, row_bytes, xcd);
    }
    return check_hip(hipGetLastError(), "calib_fill_ex launch");
}

// ---- the plain grid-stride fill (rounds 1-2's write ceiling; tools/sweep_osc.py) ---------------------------------------------
namespace mxg {
namespace {
__global__ void calib_fill8(double *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 1.0;
}
__global__ void calib_fill16(double2v *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double2v v = {1.0, 2.0};
    for (; i < n; i += stride) p[i] = v;
}
}  // namespace
}  // namespace mxg

extern "C" int mxg_calib_fill(void *d_dst, size_t bytes, int width, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_dst && (width == 8 || width == 16), "bad argument");
    hipStream_t st = resolve_stream(stream);
    if (width == 8)
        hipLaunchKernelGGL(calib_fill8, dim3(2048), dim3(256), 0, st, (double *)d_dst, bytes / 8);
    else
        hipLaunchKernelGGL(calib_fill16, dim3(2048), dim3(256), 0, st, (double2v *)d_dst,
                           bytes / 16);
    return check_hip(hipGetLastError(), "calib_fill launch");
}

