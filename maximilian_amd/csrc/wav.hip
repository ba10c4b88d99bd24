// wav.hip -- maxiSample::load / read / save: the 16-bit PCM WAV format either side of the sample banks.
//
// Reference: src/maximilian.cpp load C:605-609, read C:612-692, save C:694-725.  The file is parsed
// on the host exactly the way read() walks it (ChunkSize at byte 4, SubChunk1Size at 16, the six fmt
// fields read sequentially from byte 20 whatever SubChunk1Size says, chunks walked from
// 20+SubChunk1Size until "data"); the PCM payload goes to the device as int16 (2 B/sample over PCIe
// instead of 8) and is de-interleaved and normalised there: amplitudes[i] = short/32767.0 (C:679), an
// IEEE division, bit-exact.  save() is the inverse: static_cast<short>(round(a*32767.0)) (C:703) on the
// device, 2 B/sample back, 44-byte header from the stored fields.
//
// Multi-channel quirk kept (C:667-674): `for (i = readChannel*2; i < myDataSize+6; i += myChannels*2)
// shortAmps[position++] = shortAmps[i]` indexes SHORTS with a byte stride and a byte bound: it keeps
// every (2*channels)-th short, leaves the vector at full length (the tail keeps the interleaved data)
// and reads beyond the vector for i >= size -- undefined in the reference, 0 here.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "mxg_common.h"
#include "mxg_smp.h"

namespace mxg {
namespace {

__global__ void wav_to_amplitudes_kernel(const int16_t *__restrict__ raw, size_t n, int channels, int channel,
                                         long long dataSize, double *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // number of iterations of the reference's de-interleave loop
    long long picks = 0;
    if (channels > 1) {
        const long long first = (long long)channel * 2, bound = dataSize + 6, step = (long long)channels * 2;
        picks = first < bound ? (bound - first + step - 1) / step : 0;
    }
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        int16_t v;
        if ((long long)p < picks) {
            const size_t src = (size_t)channel * 2 + p * (size_t)channels * 2;
            v = src < n ? raw[src] : (int16_t)0;
        } else {
            v = raw[p];
        }
        out[p] = v / 32767.0;
    }
}

__global__ void amplitudes_to_wav_kernel(const double *__restrict__ amp, size_t n, int16_t *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double r = round(amp[i] * 32767.0);
        // static_cast<short>(double) on x86-64 = cvttsd2si to int32 (0x80000000 when out of range or
        // NaN), then the low 16 bits
        const int32_t w = (r >= -2147483648.0 && r < 2147483648.0) ? (int32_t)r : INT32_MIN;
        out[i] = (int16_t)(w & 0xffff);
    }
}

constexpr size_t kWavTruncationSlack = (size_t)16 << 20;  // bytes a data chunk may declare beyond the file's end

unsigned blocks_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b ? b : 1));
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

double *mxg_sample_load_wav(const char *path, int channel, size_t *h_len, int32_t *h_hdr) {
    if (ensure_init_only()) return nullptr;
    if (!path || !h_len) {
        fail(MXG_ERR_INVALID, "mxg_sample_load_wav: null argument");
        return nullptr;
    }
    FILE *f = fopen(path, "rb");
    if (!f) {  // the reference returns false and prints "ERROR: Could not load sample." (C:686)
        fail(MXG_ERR_INVALID, "mxg_sample_load_wav: cannot open %s", path);
        return nullptr;
    }
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b((size_t)(fsize > 0 ? fsize : 0));
    const bool read_ok = fsize >= 44 && fread(b.data(), 1, (size_t)fsize, f) == (size_t)fsize;
    fclose(f);
    if (!read_ok) {
        fail(MXG_ERR_INVALID, "mxg_sample_load_wav: %s is shorter than a WAV header", path);
        return nullptr;
    }
    int32_t chunkSize, subChunk1Size, sampleRate, byteRate, dataSize = 0;
    int16_t format, channels, blockAlign, bitsPerSample;
    memcpy(&chunkSize, &b[4], 4);       // C:623-624
    memcpy(&subChunk1Size, &b[16], 4);  // C:626-627
    memcpy(&format, &b[20], 2);
    memcpy(&channels, &b[22], 2);
    memcpy(&sampleRate, &b[24], 4);
    memcpy(&byteRate, &b[28], 4);
    memcpy(&blockAlign, &b[32], 2);
    memcpy(&bitsPerSample, &b[34], 2);
    long filePos = 20 + (long)subChunk1Size;  // C:650
    for (;;) {                                // C:651-662
        if (filePos < 0 || filePos + 8 > fsize) {
            fail(MXG_ERR_INVALID, "mxg_sample_load_wav: no data chunk in %s", path);
            return nullptr;
        }
        memcpy(&dataSize, &b[(size_t)filePos + 4], 4);
        const bool isdata = memcmp(&b[(size_t)filePos], "data", 4) == 0;
        filePos += 8;
        if (isdata) break;
        // The reference's walk also stops on inFile.eof() (C:651); here every skipped chunk must move the
        // cursor forward and stay inside the file, so a malformed size field cannot loop or walk backwards.
        if (dataSize < 0 || (long)dataSize > fsize - filePos) {
            fail(MXG_ERR_INVALID, "mxg_sample_load_wav: chunk size %d at byte %ld leaves %s", (int)dataSize,
                 filePos - 8, path);
            return nullptr;
        }
        filePos += dataSize;
    }
    if (dataSize < 0 || channel < 0) {
        fail(MXG_ERR_INVALID, "mxg_sample_load_wav: bad data size / channel");
        return nullptr;
    }
    size_t avail = (size_t)(fsize - filePos);
    // a truncated file leaves zeros, like the resized vector (shortAmps.resize(myDataSize/2), C:665) -- but a
    // declared size far beyond the bytes that exist is a corrupt header, not a truncation: do not allocate for it
    if ((size_t)dataSize > avail + kWavTruncationSlack) {
        fail(MXG_ERR_INVALID, "mxg_sample_load_wav: data chunk declares %d bytes, %zu present in %s", (int)dataSize,
             avail, path);
        return nullptr;
    }
    const size_t n = (size_t)dataSize / 2;
    std::vector<int16_t> sh(n ? n : 1, 0);
    if (avail > 2 * n) avail = 2 * n;
    memcpy(sh.data(), &b[(size_t)filePos], avail);

    double *base = nullptr;
    int16_t *d_raw = nullptr;
    const size_t total = n + kSmpGuardLo + kSmpGuardHi;  // same layout as mxg_sample_upload (mxg_smp.h)
    if (check_hip(hipMalloc(&base, total * sizeof(double)), "hipMalloc(sample)")) return nullptr;
    auto release = [&]() {  // every error path below gives the device buffers back
        if (d_raw) (void)hipFree(d_raw);
        (void)hipFree(base);
        return (double *)nullptr;
    };
    if (check_hip(hipMemset(base, 0, total * sizeof(double)), "hipMemset(sample)")) return release();
    if (n) {
        if (check_hip(hipMalloc(&d_raw, n * sizeof(int16_t)), "hipMalloc(wav)")) return release();
        if (check_hip(hipMemcpy(d_raw, sh.data(), n * sizeof(int16_t), hipMemcpyHostToDevice), "hipMemcpy(wav)"))
            return release();
        hipLaunchKernelGGL(wav_to_amplitudes_kernel, dim3(blocks_for(n)), dim3(256), 0, resolve_stream(nullptr), d_raw,
                           n, (int)channels, channel, (long long)dataSize, base + kSmpGuardLo);
        if (check_hip(hipGetLastError(), "wav_to_amplitudes launch")) return release();
        if (check_hip(hipStreamSynchronize(resolve_stream(nullptr)), "wav_to_amplitudes")) return release();
        (void)hipFree(d_raw);
        d_raw = nullptr;
    }
    *h_len = n;
    if (h_hdr) {
        h_hdr[0] = chunkSize; h_hdr[1] = subChunk1Size; h_hdr[2] = format; h_hdr[3] = channels;
        h_hdr[4] = sampleRate; h_hdr[5] = byteRate; h_hdr[6] = blockAlign; h_hdr[7] = bitsPerSample;
    }
    return base + kSmpGuardLo;
}

int mxg_sample_save_wav(const char *path, const double *d_samples, size_t len, const int32_t *h_hdr, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(path && h_hdr && (d_samples || len == 0), "null argument");
    MXG_REQUIRE(len <= (size_t)0x3fffffff, "too many samples for a 32-bit data chunk");
    std::vector<int16_t> sh(len ? len : 1);
    if (len) {
        hipStream_t st = resolve_stream(stream);
        int16_t *d_q = nullptr;
        MXG_HIP(hipMalloc(&d_q, len * sizeof(int16_t)));
        hipLaunchKernelGGL(amplitudes_to_wav_kernel, dim3(blocks_for(len)), dim3(256), 0, st, d_samples, len, d_q);
        MXG_HIP(hipGetLastError());
        MXG_HIP(hipMemcpyAsync(sh.data(), d_q, len * sizeof(int16_t), hipMemcpyDeviceToHost, st));
        MXG_HIP(hipStreamSynchronize(st));
        MXG_HIP(hipFree(d_q));
    }
    FILE *f = fopen(path, "wb");
    if (!f) return fail(MXG_ERR_INVALID, "mxg_sample_save_wav: cannot create %s", path);
    const int32_t chunk = h_hdr[0], sub1 = h_hdr[1], rate = h_hdr[4], brate = h_hdr[5], dsize = (int32_t)(len * 2);
    const int16_t fmt = (int16_t)h_hdr[2], ch = (int16_t)h_hdr[3], align = (int16_t)h_hdr[6], bits = (int16_t)h_hdr[7];
    bool ok = fwrite("RIFF", 1, 4, f) == 4;  // C:707-723
    ok = ok && fwrite(&chunk, 4, 1, f) == 1 && fwrite("WAVE", 1, 4, f) == 4 && fwrite("fmt ", 1, 4, f) == 4;
    ok = ok && fwrite(&sub1, 4, 1, f) == 1 && fwrite(&fmt, 2, 1, f) == 1 && fwrite(&ch, 2, 1, f) == 1;
    ok = ok && fwrite(&rate, 4, 1, f) == 1 && fwrite(&brate, 4, 1, f) == 1 && fwrite(&align, 2, 1, f) == 1;
    ok = ok && fwrite(&bits, 2, 1, f) == 1 && fwrite("data", 1, 4, f) == 4 && fwrite(&dsize, 4, 1, f) == 1;
    ok = ok && (len == 0 || fwrite(sh.data(), 2, len, f) == len);
    ok = (fclose(f) == 0) && ok;
    return ok ? MXG_OK : fail(MXG_ERR_INVALID, "mxg_sample_save_wav: short write to %s", path);
}

}  // extern "C"
