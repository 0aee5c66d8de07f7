// Host emulation of the fused 4096-channel kernel's phases — TEST INFRASTRUCTURE ONLY.
// Compiles effex_amd/csrc/fx_fused4096.h with g++ and runs the 512 "threads" of one workgroup
// phase by phase (a barrier = the end of a loop over threads), so the index logic, LDS layouts and
// the FFT decomposition can be checked against the oracle on a machine without a GPU.  Nothing
// here is linked into libfxcorr.so.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../effex_amd/csrc/fx_fused4096.h"

using namespace fxc;
using namespace fxc::fused;

extern "C" int emul_fused4096(const float* x, int64_t num_samp, const double* window, double* out_sum) {
    const int64_t P = num_samp / kN;
    std::vector<unsigned char> lds(kLdsBytes);
    f4* win = reinterpret_cast<f4*>(lds.data() + kLdsWin);
    cf* region = reinterpret_cast<cf*>(lds.data() + kLdsRegion);
    cf* tw2 = reinterpret_cast<cf*>(lds.data() + kLdsTw2);
    for (int r = 0; r < 16; ++r)
        for (int j = 0; j < 256; ++j) {
            const int m = j + 256 * r;
            f4 w;
            w.x = (float)window[0 * kN + m];
            w.y = (float)window[1 * kN + m];
            w.z = (float)window[2 * kN + m];
            w.w = (float)window[3 * kN + m];
            win[r * 256 + j] = w;
        }
    const double two_pi = 6.283185307179586476925286766559;
    for (int q1 = 0; q1 < 16; ++q1)
        for (int j0 = 0; j0 < 16; ++j0) {
            const double ph = two_pi * (double)(j0 * q1) / 256.0;
            tw2[q1 * 16 + j0] = mk((float)std::cos(ph), (float)std::sin(ph));
        }
    std::vector<State> st(kThreads);
    std::vector<cf> vbuf(kThreads * 16);
    std::vector<cf> tw1(16 * 256);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int j = 0; j < 256; ++j) {
            const double ph = two_pi * (double)((j * k1) % kN) / (double)kN;
            tw1[k1 * 256 + j] = mk((float)std::cos(ph), (float)std::sin(ph));
        }
    for (int tid = 0; tid < kThreads; ++tid) {
        state_reset_all(st[tid]);
        state_load_twiddles(st[tid], tw1.data(), tid);
    }
    const cf* xc = reinterpret_cast<const cf*>(x);
    for (int64_t i = 0; i < P; ++i) {
        // phase 1: frame i goes to ring slot i & 3 (the kernel's buffer loads), then FIR + radix-16 + twiddle
        for (int tid = 0; tid < kThreads; ++tid) {
            const int ant = tid >> 8, j = tid & 255;
            for (int r = 0; r < 16; ++r)
                st[tid].h[i & 3][r] = xc[ant * num_samp + i * kN + sample_offset(j, r)];
            cf(&v)[16] = *reinterpret_cast<cf(*)[16]>(&vbuf[tid * 16]);
            switch (i & 3) {
                case 0: phase1_fir<0>(st[tid], win, tid, v); break;
                case 1: phase1_fir<1>(st[tid], win, tid, v); break;
                case 2: phase1_fir<2>(st[tid], win, tid, v); break;
                default: phase1_fir<3>(st[tid], win, tid, v); break;
            }
            dft16_a(v);
        }
        for (int tid = 0; tid < kThreads; ++tid)
            phase1_finish_store(st[tid], *reinterpret_cast<cf(*)[16]>(&vbuf[tid * 16]), region, tid);
        // barrier; phase 2 (reads complete for a whole wave before its stores: emulate per wave)
        for (int wave = 0; wave < kThreads / 64; ++wave) {
            for (int l = 0; l < 64; ++l) {
                const int tid = wave * 64 + l;
                cf(&v)[16] = *reinterpret_cast<cf(*)[16]>(&vbuf[tid * 16]);
                phase2_load(region, tid, v);
                dft16(v);
                phase2_twiddle(v, tw2, tid);
            }
            for (int l = 0; l < 64; ++l) {
                const int tid = wave * 64 + l;
                phase2_store(*reinterpret_cast<cf(*)[16]>(&vbuf[tid * 16]), region, tid);
            }
            for (int l = 0; l < 64; ++l) {
                const int tid = wave * 64 + l;
                cf(&v)[16] = *reinterpret_cast<cf(*)[16]>(&vbuf[tid * 16]);
                phase3_load(region, tid, v);
                dft16(v);
            }
            // permlane32_swap pairing: lane l < 32 holds antenna 0, lane l + 32 antenna 1
            for (int l = 0; l < 32; ++l) {
                const int lo = wave * 64 + l, hi = lo + 32;
                const cf* vlo = &vbuf[lo * 16];
                const cf* vhi = &vbuf[hi * 16];
                for (int q = 0; q < 8; ++q) {
                    xacc(st[lo], q, vlo[q], vhi[q]);          // A = [A.lo | B.lo], B = [A.hi | B.hi]
                    xacc(st[hi], q, vlo[q + 8], vhi[q + 8]);
                }
            }
        }
    }
    for (int tid = 0; tid < kThreads; ++tid)
        for (int q = 0; q < kAccPerThread; ++q) {
            const int k = bin_of(tid, q);
            if (slot_of_bin(k) != q * kThreads + tid) return -1;
            out_sum[2 * k] = st[tid].acc[q].x;
            out_sum[2 * k + 1] = st[tid].acc[q].y;
        }
    return 0;
}

// layout self-checks: the slot order of the raw sums and the spectrum order of the F-only variant
extern "C" int emul_layout_check(void) {
    for (int tid = 0; tid < kThreads; ++tid) {
        const int l = tid & 63, wave = tid >> 6;
        const int k1 = 2 * wave + ((l >> 4) & 1), q1 = l & 15;
        for (int q2 = 0; q2 < 16; ++q2) {
            const int k = k1 + 16 * q1 + 256 * q2;
            if (specpos_of_bin(k) != specpos(lane_specpos(tid), q2)) return -1;
        }
        for (int q = 0; q < kAccPerThread; ++q)
            if (slot_of_bin(bin_of(tid, q)) != q * kThreads + tid) return -2;
    }
    return 0;
}

// 16-point DFT self-check hook
extern "C" void emul_dft16(const float* in, float* out) {
    cf v[16];
    for (int n = 0; n < 16; ++n) v[n] = mk(in[2 * n], in[2 * n + 1]);
    dft16(v);
    for (int n = 0; n < 16; ++n) { out[2 * n] = v[n].x; out[2 * n + 1] = v[n].y; }
}
