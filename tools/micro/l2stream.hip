// How fast can ONE 16-wave workgroup per CU stream a weight array that every CU reads (L2-resident, as the GEMM stages of
// the attention kernel do), as a function of the fragments each wave keeps in flight?  256 workgroups x 16 waves; every
// wave reads its own sixteenth of a 336 KB array as 1 KB fragments (global_load_dwordx4 per lane), D fragments requested
// ahead of the one consumed (the order pinned by sched_barrier, as in WaveGemm), PASSES times over.  MF = 0: a fragment is
// consumed by four adds; MF = 1: by four dependent 16x16x4 fp32 MFMAs (the folded tail's chain); MF = 2: by four MFMAs
// into two accumulators; MF = 3: by one add, with four MFMAs on register operands beside it (do the loads cost the matrix
// pipe time even when it does not depend on them?).  Prints the in-kernel clock too (shader cycles over the 100 MHz counter).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int FRAGS = 384, PASSES = 64, FW = 24;      // 384 KB; 24 fragments per wave and pass
template <int D, int MF, int ROT = 0, int SADDR = 0, int CONT = 0, int WIDTH = 4>
__global__ __launch_bounds__(1024) void k_stream(const f32x4* __restrict__ w, float* out, long long* clk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (MF = 5: six 1 KB fragments per wave, read round and round; MF = 6: each wave's ring of D + 1 fragments)
  __shared__ f32x4 lds4[MF == 5 ? 16 * 6 * 64 : MF == 6 ? 16 * (D + 1) * 64 : 64];
  if (MF == 5) { for (int i = threadIdx.x; i < 16 * 6 * 64; i += 1024) lds4[i] = f32x4{1.f, 2.f, 3.f, 4.f}; __syncthreads(); }
  f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const float av = (float)lane;
  // CONT = 1: the ring is kept full across the passes (the refill of the last D steps of a pass are the first fragments of
  // the next; FW is a multiple of D + 1, so the slot of every step is known at compile time) -- the steady state, without
  // the empty pipe a pass otherwise starts from
  constexpr int DD = D ? D : 1;             // D = 0: no loads at all, the fragments are a register constant
  static_assert(!CONT || FW % (DD + 1) == 0, "ring slots must repeat per pass");
  f32x4 r[DD + 1];                          // D requested ahead + the one being consumed
  for (int p = 0; p < PASSES; ++p) {
    const f32x4* base = w + (((wave + (ROT ? blockIdx.x : 0)) & 15) * FW) * 64 + lane;
    int lo = lane;
    asm volatile("" : "+v"(lo));            // (keeps the compiler from hoisting the loads out of the pass loop)
    base += lo - lane;
    const char* sbase = (const char*)w + __builtin_amdgcn_readfirstlane(((wave + (ROT ? blockIdx.x : 0)) & 15) * FW * 1024);
    const unsigned voff = (unsigned)lo * 16u;
    auto frag = [&](int i) -> f32x4 {
      if (SADDR) return *reinterpret_cast<const f32x4*>(sbase + i * 1024 + voff);
      if (WIDTH == 1) {                     // as many load instructions, a quarter of the bytes: is the cost per instruction?
        const float x = *reinterpret_cast<const float*>(&base[i * 64]);
        return f32x4{x, x, x, x};
      }
      if (WIDTH == 8) {                     // twice the instructions for twice the bytes
        const f32x4 x = base[i * 64], y = base[((i + 12) % FW) * 64];
        return x + y;
      }
      return base[i * 64];
    };
    // MF = 6: the fragment goes from L2 straight into LDS (global_load_lds_dwordx4: no register is written) and is read
    // from there with ds_read_b128 when its turn comes
    auto frag_lds = [&](int i, int slot) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)&base[i * 64],
                                       (__attribute__((address_space(3))) void*)&lds4[(wave * (DD + 1) + slot) * 64], 16, 0, 0);
    };
    if (!CONT || p == 0) {
#pragma unroll
      for (int i = 0; i < DD; ++i) {
        if (MF == 6) frag_lds(i, i);
        else r[i] = D ? frag(i) : f32x4{av, av, av, av};
      }
    }
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      if (MF == 6) {
        if (CONT || f + D < FW) frag_lds((f + D) % FW, (f + D) % (DD + 1));
        __builtin_amdgcn_sched_barrier(0);
        f32x4 l;
        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&lds4[(wave * (DD + 1) + f % (DD + 1)) * 64 + lane];
        // (hand-placed waits: D loads may still be in flight behind this fragment's)
        if (CONT || f + D < FW) __builtin_amdgcn_s_waitcnt(0x0F70 | (D & 15) | ((D >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(l) : "v"(la) : "memory");
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, l[s], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (D && (CONT || f + D < FW)) r[(f + D) % (DD + 1)] = frag((f + D) % FW);   // the refill goes out ahead of this step's MFMAs
      const f32x4 cur = r[D ? f % (DD + 1) : 0];
      __builtin_amdgcn_sched_barrier(0);
      if (MF == 7) {                        // 64 FMAs on eight accumulators per fragment (what does a load cost a VALU-bound wave?)
        float x0 = acc[0], x1 = acc[1], x2 = acc[2], x3 = acc[3], x4 = acc2[0], x5 = acc2[1], x6 = acc2[2], x7 = acc2[3];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const float c = cur[s & 3];
          x0 = __builtin_fmaf(x0, av, c); x1 = __builtin_fmaf(x1, av, c); x2 = __builtin_fmaf(x2, av, c); x3 = __builtin_fmaf(x3, av, c);
          x4 = __builtin_fmaf(x4, av, c); x5 = __builtin_fmaf(x5, av, c); x6 = __builtin_fmaf(x6, av, c); x7 = __builtin_fmaf(x7, av, c);
        }
        acc = f32x4{x0, x1, x2, x3}; acc2 = f32x4{x4, x5, x6, x7};
      } else if (MF == 5) {                        // (D = 0) one ds_read_b128 per four MFMAs supplies the B operand from LDS
        const f32x4 l = lds4[(wave * 6 + f % 6) * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, l[s], acc, 0, 0, 0);
      } else if (MF == 0) acc += cur;
      else if (MF == 3) {                   // the loads are waited for at the same places, the MFMAs do not use them
        acc2 += cur;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, av, acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (MF == 2 && (s & 1)) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cur[s], acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cur[s], acc, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  acc += acc2;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
  if (threadIdx.x == 0) {
    clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0;
    clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}
template <int D, int MF, int ROT = 0, int SADDR = 0, int CONT = 0, int WIDTH = 4> void run(const f32x4* w, float* out, long long* clk) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 20; ++rep) {
    (void)hipEventRecord(e0);
    k_stream<D, MF, ROT, SADDR, CONT, WIDTH><<<256, 1024>>>(w, out, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  long long h[512];
  (void)hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
  // (the stamps are workgroup 0's first wave's: good for the clock, not for the span -- the oldest wave of a SIMD is served first)
  const double mhz = (double)h[0] / ((double)h[1] / 100.0), cyc = ms * 1e3 * mhz;
  const double bytes = D ? FRAGS * 1024.0 * PASSES : 0.0;
  printf("%s%s%s%sMF=%d D=%2d: %7.1f us  %5.1f B/clk/CU  %5.1f GB/s per CU  %4.1f TB/s chip  clock %4.0f MHz", ROT ? "rotated " : "", SADDR ? "saddr " : "", CONT ? "continuous " : "", WIDTH == 1 ? "dword " : WIDTH == 8 ? "two-loads " : "", MF, D, ms * 1e3, bytes / cyc,
         bytes / (ms * 1e-3) / 1e9, 256.0 * bytes / (ms * 1e-3) / 1e12, mhz);
  if (MF == 7) printf("  %5.1f cycles per fragment (64 FMAs) and SIMD", cyc / (4.0 * FW * PASSES));
  else if (MF) printf("  %5.1f cycles per MFMA and SIMD", cyc / (4.0 * 4 * FW * PASSES));
  printf("\n");
}
int main() {
  f32x4* w; float* out; long long* clk;
  (void)hipMalloc(&w, FRAGS * 1024);
  (void)hipMemset(w, 0, FRAGS * 1024);
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&clk, 512 * 8);
  run<1, 0>(w, out, clk); run<2, 0>(w, out, clk); run<3, 0>(w, out, clk); run<7, 0>(w, out, clk);
  run<0, 1>(w, out, clk); run<0, 2>(w, out, clk);
  run<1, 1>(w, out, clk); run<2, 1>(w, out, clk); run<3, 1>(w, out, clk); run<5, 1>(w, out, clk); run<7, 1>(w, out, clk);
  run<1, 1, 0, 0, 1>(w, out, clk); run<2, 1, 0, 0, 1>(w, out, clk); run<3, 1, 0, 0, 1>(w, out, clk); run<5, 1, 0, 0, 1>(w, out, clk);
  run<7, 1, 0, 0, 1>(w, out, clk); run<11, 1, 0, 0, 1>(w, out, clk);
  run<2, 2, 0, 0, 1>(w, out, clk); run<7, 2, 0, 0, 1>(w, out, clk);
  run<2, 3, 0, 0, 1>(w, out, clk);
  run<2, 1, 0, 0, 1, 1>(w, out, clk); run<3, 1, 0, 0, 1, 1>(w, out, clk); run<2, 1, 0, 0, 1, 8>(w, out, clk); run<3, 1, 0, 0, 1, 8>(w, out, clk);
  run<0, 5>(w, out, clk);
  run<0, 7>(w, out, clk); run<2, 7, 0, 0, 1>(w, out, clk); run<2, 7, 0, 0, 1, 1>(w, out, clk); run<2, 7, 0, 0, 1, 8>(w, out, clk);
  run<1, 6, 0, 0, 1>(w, out, clk); run<2, 6, 0, 0, 1>(w, out, clk); run<3, 6, 0, 0, 1>(w, out, clk); run<5, 6, 0, 0, 1>(w, out, clk);
  run<2, 1, 1, 0, 1>(w, out, clk); run<2, 1, 0, 1, 1>(w, out, clk);
  return 0;
}
