// Micro-benchmark: store throughput of the matrix-core epilogue's access pattern.
// A workgroup (4 waves) writes a [32 rows][N] u16 block (row pitch 2N bytes, N = 821) strip by strip, like k_encrypt_m:
//   pattern 0: one instruction = lanes 0-31 -> 32 consecutive columns of row r, lanes 32-63 -> the same columns of row r+4
//              (the accumulator layout of v_mfma_i32_32x32x32_i8)
//   pattern 1: one instruction = 64 consecutive columns of ONE row (after a v_permlane32_swap between two tiles)
//   pattern 3: a strip's row segment (192-256 bytes at a 2-byte aligned start) as ALIGNED 16-byte pieces (lane = piece,
//              4 rows x 16 pieces per instruction) plus its two edges as 2-byte stores (lane = row x element): 4 store
//              instructions per 8 rows and strip instead of 12-16
//   pattern 4: the whole row block as one contiguous run of aligned 16-byte pieces (what a full LDS image would allow)
// hipcc -O3 --offload-arch=gfx950 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int PAT>
__global__ __launch_bounds__(256) void k(unsigned short* out, long nrb, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    unsigned short* blk = out + rb * 32 * N;
    if (PAT == 4) {
      uint4* b4 = (uint4*)blk;                                   // 32 N u16 = 64 N bytes: a multiple of 16
      for (int i = threadIdx.x; i < 4 * N; i += 256) b4[i] = make_uint4((unsigned)rb, i, i, i);
      continue;
    }
    for (int strip = wave; strip < 8; strip += 4) {           // 8 strips of 3-4 tiles; here: 3 tiles of 32 columns (+1 for even strips)
      const int kb0 = strip * 3 + (strip < 2 ? strip : 2), nt = strip < 2 ? 4 : 3;
      for (int j = 0; j < 4; j++)
        for (int ii = 0; ii < 4; ii++) {
          const int ro = ii + 8 * j;
          if (PAT == 0) {
            for (int t = 0; t < nt; t++) {
              const int col = 32 * (kb0 + t) + (lane & 31), row = ro + 4 * (lane >> 5);
              if (col < N) blk[row * N + col] = (unsigned short)(rb + col);
            }
          } else if (PAT == 3) {
            if (ii == 0 && 32 * kb0 < N) {                       // (strips past the row end of a smaller N: nothing to write)
              const int c0 = 32 * kb0, c1 = 32 * (kb0 + nt) < N ? 32 * (kb0 + nt) : N;
              for (int half = 0; half < 2; half++) {
                const int row = 8 * j + 4 * half + (lane >> 4);
                const unsigned long long S = (unsigned long long)(blk + row * N + c0), E = (unsigned long long)(blk + row * N + c1);
                const unsigned long long A0 = (S + 15) & ~15ull, A1 = E & ~15ull, pp = A0 + 16 * (lane & 15);
                if (pp + 16 <= A1) *(uint4*)pp = make_uint4((unsigned)rb, lane, row, c0);
              }
              for (int side = 0; side < 2; side++) {
                const int row = 8 * j + (lane >> 3), el = lane & 7;
                const unsigned long long S = (unsigned long long)(blk + row * N + c0), E = (unsigned long long)(blk + row * N + c1);
                const unsigned long long A0 = (S + 15) & ~15ull, A1 = E & ~15ull;
                const unsigned long long q = (side == 0 ? S : A1) + 2 * el;
                if (side == 0 ? q < A0 : q < E) *(unsigned short*)q = (unsigned short)(rb + el);
              }
            }
          } else if (PAT == 2) {                                 // quad-packed: one 8-byte store per lane = 4 columns of row (lane & 3)
            if (ii == 0)
              for (int t = 0; t < nt; t++) {
                const int col = 32 * (kb0 + t) + (lane & 28), row = 8 * j + (lane & 3) + 4 * (lane >> 5);
                if (col + 4 <= N) *(unsigned long long*)(blk + row * N + col) = (unsigned long long)(rb + col) * 0x0001000100010001ull;
              }
          } else {
            for (int half = 0; half < 2; half++)                 // rows ro and ro + 4, each as 64-column instructions
              for (int t = 0; t < nt; t += 2) {
                const int col = 32 * (kb0 + t) + lane, row = ro + 4 * half;
                if (col < N && (t + 1 < nt || lane < 32)) blk[row * N + col] = (unsigned short)(rb + col);
              }
          }
        }
    }
  }
}

// pattern 5: pattern 1 on rows at a pitch of 832 elements (1664 bytes = 13 cache lines): every instruction is one whole line
template <int PAT>
__global__ __launch_bounds__(256) void k5(unsigned short* out, long nrb, int N, int LD) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    unsigned short* blk = out + rb * 32 * LD;
    for (int strip = wave; strip < 8; strip += 4) {
      const int kb0 = strip * 3 + (strip < 2 ? strip : 2), nt = strip < 2 ? 4 : 3;
      for (int ro = 0; ro < 32; ro++)
        for (int t = 0; t < nt; t += 2) {
          const int col = 32 * (kb0 + t) + lane;
          if (col < N && (t + 1 < nt || lane < 32)) blk[ro * LD + col] = (unsigned short)(rb + col);
        }
    }
  }
}

// patterns 6 / 7: the TRANSPOSED accumulator layout (matrix instruction with its operands swapped: lane & 31 = row of the row block,
// registers = columns (i & 3) + 8 (i >> 2) + 4 (lane >> 5) of the tile).  A lane owns 4 consecutive columns per register group;
// 6: after one v_permlane32_swap per dword the lower half-wave holds columns 0-7 / 16-23 of its row and the upper half columns
//    8-15 / 24-31: one 16-byte store per lane, 32 rows x 32 contiguous bytes per instruction (2-byte aligned addresses);
// 7: without the swap: one 8-byte store per lane, 32 rows x 2 pieces of 8 bytes per instruction.
template <int PAT>
__global__ __launch_bounds__(256) void k6(unsigned short* out, long nrb, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  for (long rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    unsigned short* row = out + (rb * 32 + r) * N;
    for (int strip = wave; strip < 8; strip += 4) {
      const int kb0 = strip * 3 + (strip < 2 ? strip : 2), nt = strip < 2 ? 4 : 3;
      for (int t = 0; t < nt; t++) {
        if (PAT == 6) {
          for (int g2 = 0; g2 < 2; g2++) {
            const int col = 32 * (kb0 + t) + 16 * g2 + 8 * h;
            if (col + 8 <= N) *(uint4*)(row + col) = make_uint4((unsigned)rb, col, r, t);
            else for (int c = col; c < N; c++) row[c] = (unsigned short)c;
          }
        } else {
          for (int k = 0; k < 4; k++) {
            const int col = 32 * (kb0 + t) + 8 * k + 4 * h;
            if (col + 4 <= N) *(uint2*)(row + col) = make_uint2((unsigned)rb, col);
            else for (int c = col; c < N; c++) row[c] = (unsigned short)c;
          }
        }
      }
    }
  }
}

// patterns 8 / 9: POLYPHASE column assignment (tile t of a group of P tiles holds the columns c0 + P l + t, l = lane & 31): a lane owns
// P adjacent columns of its row in P registers, packs them and stores 2 P bytes; one instruction = lanes 0-31 -> ONE contiguous run of
// 64 P bytes of row R, lanes 32-63 the same of row R + 4.  8: P = 2 (4 bytes per lane, 128-byte runs); 9: P = 4 (8 bytes, 256-byte runs).
// The lane that holds the row end stores its valid columns as 2-byte pieces.
template <int P>
__global__ __launch_bounds__(256) void k8(unsigned short* out, long nrb, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 31, h = lane >> 5;
  const int groups = (N + 32 * P - 1) / (32 * P);               // groups of P tiles
  for (long rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    unsigned short* blk = out + rb * 32 * N;
    for (int gi = wave; gi < groups; gi += 4) {
      const int col = 32 * P * gi + P * l;
      for (int j = 0; j < 4; j++)
        for (int ii = 0; ii < 4; ii++) {
          unsigned short* q = blk + (ii + 8 * j + 4 * h) * N + col;
          if (col + P <= N) {
            if (P == 2) *(unsigned*)q = (unsigned)(rb + col);
            else *(uint2*)q = make_uint2((unsigned)rb, col);
          } else for (int c = col; c < N; c++) q[c - col] = (unsigned short)c;
        }
    }
  }
}

// pattern 10: pattern 8 with every dword store ALIGNED: a row whose group segment starts at an address = 2 mod 4 stores the dwords
// (column 2l+1, column 2l+2) -- the second half fetched from the next lane -- for l < 31 and its two edge columns (first and last of the
// group) as one 2-byte store instruction with two active lanes per half-wave.
__global__ __launch_bounds__(256) void k10(unsigned short* out, long nrb, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 31, h = lane >> 5;
  const int groups = (N + 63) / 64;
  for (long rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    unsigned short* blk = out + rb * 32 * N;
    for (int gi = wave; gi < groups; gi += 4) {
      const int col = 64 * gi + 2 * l;
      for (int j = 0; j < 4; j++)
        for (int ii = 0; ii < 4; ii++) {
          const int row = ii + 8 * j;                                   // lanes 32-63: row + 4 (same parity)
          unsigned short* q = blk + (row + 4 * h) * N + col;
          const bool odd = (((unsigned long long)(blk + row * N + 64 * gi)) & 2) != 0;     // wave-uniform
          if (!odd) {
            if (col + 2 <= N) *(unsigned*)q = (unsigned)(rb + col);
            else if (col < N) q[0] = (unsigned short)col;
          } else {
            if (l < 31 && col + 3 <= N) *(unsigned*)(q + 1) = (unsigned)(rb + col);
            else if (l < 31 && col + 1 < N) q[1] = (unsigned short)col;
            if ((l == 0 || l == 31) && col + (l ? 1 : 0) < N) q[l ? 1 : 0] = (unsigned short)(rb);
          }
        }
    }
  }
}

static void launch(int pat, int cus, unsigned short* d, long nrb, int N) {
  if (pat == 10) { k10<<<cus * 2, 256>>>(d, nrb, N); return; }
  if (pat == 8) { k8<2><<<cus * 2, 256>>>(d, nrb, N); return; }
  if (pat == 9) { k8<4><<<cus * 2, 256>>>(d, nrb, N); return; }
  if (pat == 6) { k6<6><<<cus * 2, 256>>>(d, nrb, N); return; }
  if (pat == 7) { k6<7><<<cus * 2, 256>>>(d, nrb, N); return; }
  if (pat == 0) k<0><<<cus * 2, 256>>>(d, nrb, N); else if (pat == 1) k<1><<<cus * 2, 256>>>(d, nrb, N); else if (pat == 2) k<2><<<cus * 2, 256>>>(d, nrb, N);
  else if (pat == 3) k<3><<<cus * 2, 256>>>(d, nrb, N); else if (pat == 4) k<4><<<cus * 2, 256>>>(d, nrb, N); else k5<5><<<cus * 2, 256>>>(d, nrb, N, 832);
}
static const char* pat_name(int pat) {
  return pat == 0 ? "2 rows x 64 B per instruction" : pat == 1 ? "1 row x 128 B per instruction" : pat == 2 ? "8 rows x 64 B per instruction (8 B per lane)"
       : pat == 3 ? "aligned 16 B pieces + 2 B edges per row segment" : pat == 4 ? "whole row block, aligned 16 B pieces" : pat == 5 ? "1 row x 128 B per instruction, rows pitched to 832 elements"
       : pat == 10 ? "polyphase P = 2 with aligned dwords on odd rows (+ one 2-lane edge store)"
       : pat == 8 ? "polyphase P = 2: 2 rows x 128 B per instruction (4 B per lane, 2-byte aligned)" : pat == 9 ? "polyphase P = 4: 2 rows x 256 B per instruction (8 B per lane)"
       : pat == 6 ? "transposed layout: 32 rows x 32 B per instruction (16 B per lane, 2-byte aligned)" : "transposed layout: 32 rows x 2 x 8 B per instruction (8 B per lane)";
}

// store_pattern [N]                  -> every pattern once (GB/s)
// store_pattern N pattern seconds    -> that pattern in a loop for `seconds` (for tools/power_sample.py: J per GB)
int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  char pci[64] = ""; (void)hipDeviceGetPCIBusId(pci, sizeof pci, 0);
  printf("pci_bus_id %s\n", pci); fflush(stdout);
  const int cus = prop.multiProcessorCount; const int N = argc > 1 ? atoi(argv[1]) : 821;
  const long nrb = 32768;
  unsigned short* d; CK(hipMalloc(&d, (size_t)nrb * 32 * 832 * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (argc > 3) {
    const int pat = atoi(argv[2]); const double secs = atof(argv[3]);
    launch(pat, cus, d, nrb, N); CK(hipDeviceSynchronize());
    double total_ms = 0; long n = 0;
    while (total_ms < secs * 1e3) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 50; i++) launch(pat, cus, d, nrb, N);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      total_ms += ms; n += 50;
    }
    printf("pattern %d (%s): %ld launches, %.3f ms each, %.0f GB/s, %.3f GB per launch\n", pat, pat_name(pat), n, total_ms / n,
           (double)nrb * 32 * N * 2 / (total_ms / n) * 1e-6, (double)nrb * 32 * N * 2 * 1e-9);
    return 0;
  }
  for (int pat = 0; pat < 11; pat++)
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      launch(pat, cus, d, nrb, N);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("pattern %d (%s): %.3f ms  %.0f GB/s\n", pat, pat_name(pat), ms, (double)nrb * 32 * N * 2 / ms * 1e-6);
    }
  return 0;
}
