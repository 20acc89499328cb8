// Stand-alone probe of the decode-shaped kernels (csrc/ff_decode.hip is #included: same kernel, same launcher): correctness against a
// float64 host reference for any (M, N, K, nb, kslices, ln), cold-weight timing over rotating weight buffers, and - built with
// -DFF_DEC_TIMELINE - where a workgroup's time goes (phase timestamps per workgroup).  No Python, starts in a second:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFF_DEC_TIMELINE -I flamingo-mini_amd/csrc tools/experiments/decode_probe.hip -o /tmp/decode_probe
//     /tmp/decode_probe M N K [nb kslices ln nbuf reps]
#include <string.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../../flamingo-mini_amd/csrc/ff_decode.hip"

namespace ff {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return FF_ERR_LAUNCH; } return FF_OK; }
}

static float bf(float x) { unsigned u; memcpy(&u, &x, 4); unsigned r = ((u >> 16) & 1) + 0x7FFF; u = (u + r) & 0xFFFF0000u; float y; memcpy(&y, &u, 4); return y; }
static unsigned short bfbits(float x) { float y = bf(x); unsigned u; memcpy(&u, &y, 4); return (unsigned short)(u >> 16); }
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

int main(int argc, char** argv) {
    using namespace ff;
    int M = argc > 1 ? atoi(argv[1]) : 32, N = argc > 2 ? atoi(argv[2]) : 5120, K = argc > 3 ? atoi(argv[3]) : 1280;
    int nb = argc > 4 ? atoi(argv[4]) : 0, ks = argc > 5 ? atoi(argv[5]) : 1, ln = argc > 6 ? atoi(argv[6]) : 1;
    int nbuf = argc > 7 ? atoi(argv[7]) : 40, reps = argc > 8 ? atoi(argv[8]) : 5;
    if (nb <= 0) nb = pick_nb(N, ks);
    const int kslice = K / ks;
    printf("M %d N %d K %d nb %d kslices %d (slice %d) ln %d: grid %d\n", M, N, K, nb, ks, kslice, ln, cdiv(N, nb) * ks);
    unsigned seed = 12345;
    std::vector<float> A((size_t)M * K), W((size_t)N * K), G(K), Bt(K), R((size_t)M * N);
    for (auto& v : A) v = bf(frand(seed) * 1.5f + 0.3f);
    for (auto& v : W) v = bf(frand(seed) * 0.05f);
    for (auto& v : G) v = bf(1.f + 0.2f * frand(seed));
    for (auto& v : Bt) v = bf(0.1f * frand(seed));
    for (auto& v : R) v = bf(frand(seed));
    auto upload = [](const std::vector<float>& h) { std::vector<unsigned short> b(h.size()); for (size_t i = 0; i < h.size(); i++) b[i] = bfbits(h[i]); void* d; hipMalloc(&d, b.size() * 2); hipMemcpy(d, b.data(), b.size() * 2, hipMemcpyHostToDevice); return (bf16*)d; };
    bf16 *dA = upload(A), *dG = upload(G), *dB = upload(Bt), *dR = upload(R);
    std::vector<bf16*> dW(nbuf);
    for (int i = 0; i < nbuf; i++) dW[i] = upload(W);
    bf16 *dC, *dAux, *dXn; float *dMean, *dRstd, *dPart; unsigned* dTick;
    hipMalloc(&dC, (size_t)M * N * 2); hipMalloc(&dAux, (size_t)M * N * 2); hipMalloc(&dXn, (size_t)M * K * 2);
    hipMalloc(&dMean, 128); hipMalloc(&dRstd, 128); hipMalloc(&dPart, (size_t)ks * 32 * N * 4); hipMalloc(&dTick, 4096);
    float gate_h = 0.5f; bf16* dGate; { std::vector<float> gv(1, gate_h); dGate = upload(gv); }
    DecodeArgs a = {};
    a.M = M; a.N = N; a.kslice = kslice; a.kslices = ks; a.nb = nb; a.ln = ln; a.act = ln ? FF_ACT_GELU : FF_ACT_NONE; a.eps = 1e-5f; a.scale = 1.f;
    a.lda = K; a.ldb = K; a.ldc = N; a.ldr = N; a.ldxn = K;
    a.A = dA; a.gamma = dG; a.beta = dB; a.C = dC; a.aux_out = dAux; a.residual = ln ? nullptr : dR; a.gate = ln ? nullptr : dGate;
    a.xn_out = dXn; a.mean = dMean; a.rstd = dRstd; a.partial = dPart; a.tickets = dTick;
    hipStream_t st; hipStreamCreate(&st);
    // ---- correctness ----
    a.B = dW[0];
    hipMemsetAsync(dTick, 0, 4096, st);
    if (launch_decode(a, st) != FF_OK) return 1;
    hipStreamSynchronize(st);
    std::vector<unsigned short> hc((size_t)M * N);
    hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost);
    double err2 = 0, ref2 = 0;
    std::vector<float> X = A;
    if (ln) for (int m = 0; m < M; m++) { double mu = 0, sq = 0; for (int k = 0; k < K; k++) mu += A[(size_t)m * K + k]; mu /= K; for (int k = 0; k < K; k++) sq += (A[(size_t)m * K + k] - mu) * (A[(size_t)m * K + k] - mu);
        double rs = 1.0 / sqrt(sq / K + 1e-5); for (int k = 0; k < K; k++) X[(size_t)m * K + k] = bf((float)((A[(size_t)m * K + k] - mu) * rs * G[k] + Bt[k])); }
    for (int m = 0; m < M; m++) for (int n = 0; n < N; n++) {
        double acc = 0; for (int k = 0; k < K; k++) acc += (double)X[(size_t)m * K + k] * W[(size_t)n * K + k];
        double v = acc;
        if (ln) v = 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
        else v = v * tanh((double)gate_h) + R[(size_t)m * N + n];
        unsigned u = (unsigned)hc[(size_t)m * N + n] << 16; float got; memcpy(&got, &u, 4);
        err2 += (got - v) * (got - v); ref2 += v * v;
    }
    printf("relative L2 error vs float64 host reference: %.3e\n", sqrt(err2 / ref2));
    // ---- timing: rotating (cold) weights ----
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < nbuf; i++) { a.B = dW[i]; hipMemsetAsync(dTick, 0, 4096, st); launch_decode(a, st); }
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; r++) for (int i = 0; i < nbuf; i++) { a.B = dW[i]; if (ks > 1) hipMemsetAsync(dTick, 0, 4096, st); launch_decode(a, st); }
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / (reps * nbuf);
    printf("%.2f us per launch (eager, incl. gaps%s), weights %.1f MB -> %.2f TB/s\n", us, ks > 1 ? " and a memset" : "", (double)N * K * 2 / 1e6, (double)N * K * 2 / us / 1e6);
#ifdef FF_DEC_TIMELINE
    a.B = dW[nbuf / 2]; hipMemsetAsync(dTick, 0, 4096, st); launch_decode(a, st); hipStreamSynchronize(st);
    const int grid = cdiv(N, nb) * ks;
    std::vector<unsigned long long> tl((size_t)1024 * 8);
    hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(ff::g_dec_timeline), tl.size() * 8);
    unsigned long long t0 = ~0ull, t_end = 0; double ph[7] = {0, 0, 0, 0, 0, 0, 0};
    const int nblk = grid < 1024 ? grid : 1024;
    for (int b = 0; b < nblk; b++) { if (tl[b * 8] < t0) t0 = tl[b * 8]; if (tl[b * 8 + 5] > t_end) t_end = tl[b * 8 + 5]; }
    for (int b = 0; b < nblk; b++) for (int i = 0; i < 6; i++) ph[i] += (double)(tl[b * 8 + i] - t0) / nblk;
    printf("timeline (us after the first workgroup's start, mean over %d workgroups; 10 ns ticks): start %.2f | loads issued %.2f | rows landed %.2f | LN done %.2f | "
           "MFMA done (weights landed) %.2f | stored %.2f ;  last workgroup done %.2f\n", nblk, ph[0] / 100, ph[1] / 100, ph[2] / 100, ph[3] / 100, ph[4] / 100, ph[5] / 100, (double)(t_end - t0) / 100);
#endif
    return 0;
}
