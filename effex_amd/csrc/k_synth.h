// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// synthetic IQ (effex_amd/synth.py, bit for bit)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_kernel(cf* __restrict__ x, uint64_t seed, int64_t first_chunk, int64_t n_chunks, int n_ant,
                             int64_t num_samp, const int* __restrict__ delays, const cf* __restrict__ tone,
                             int tone_period, const float* __restrict__ lut) {
    const int64_t total = n_chunks * n_ant * num_samp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint64_t key_seed = seed * 0x8CB92BA72F3D8DD7ull;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t n = idx % num_samp;
        const int64_t ca = idx / num_samp;
        const int a = (int)(ca % n_ant);
        const int64_t c = ca / n_ant;
        const uint64_t g = (uint64_t)(1 << 20) + (uint64_t)((first_chunk + c) * num_samp) + (uint64_t)n;
        const uint64_t gd = g - (uint64_t)delays[a];
        const uint64_t hs = mix64(key_seed + gd);   // stream 0 = sky
        const uint64_t hr = mix64(key_seed + (uint64_t)(a + 1) * 0xD1B54A32D192ED03ull + g);
        const cf t = tone[(int)(gd % (uint64_t)tone_period)];
        const float s_re = lut[hs & 0xFF], s_im = lut[(hs >> 8) & 0xFF];
        const float r_re = lut[hr & 0xFF], r_im = lut[(hr >> 8) & 0xFF];
        // (s + 0.5 r) + t with one rounding per step; 0.5*r is exact
        const float re = __fadd_rn(__fadd_rn(s_re, 0.5f * r_re), t.x);
        const float im = __fadd_rn(__fadd_rn(s_im, 0.5f * r_im), t.y);
        x[idx] = fxc::mk(re, im);
    }
}

}  // namespace
