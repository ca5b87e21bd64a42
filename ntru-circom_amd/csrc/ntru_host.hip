// ntru_host.hip -- the host-pointer entry points of include/ntru_engine.h (the ones Node.js reaches through the addon).
//
// The reference keeps everything in JS arrays (index.js:87-197); a host binding therefore hands the engine HOST buffers.
// This file moves them through the GPU as a pipeline instead of "allocate, copy, run, copy, free" per call:
//   * a batch is cut into chunks; a chunk passes three stages -- upload, compute, download -- each on its own engine-owned stream,
//     chained by events: while chunk k computes, chunk k+1 is uploaded and chunk k-1 downloaded, one copy per direction at a time
//     (PCIe is full duplex), and the CPU-side staging copies of the next chunk overlap all three;
//   * chunk k owns buffer set k % 3 (pinned host arena, device arena, scratch: they only grow -- no hipMalloc / hipFree /
//     hipHostMalloc on the steady-state path) until its download is done;
//   * buffers the caller allocated with ntru_host_alloc (pinned; the addon exposes them as TypedArrays) are DMA'd in place;
//     ordinary pageable memory is staged through the set's pinned arena with a multi-threaded memcpy.
// Shared key rows (h, f, fp) travel with every chunk (<= 4N bytes), which keeps a single-item call at one H2D, one launch
// and one D2H with a single host synchronisation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "engine_internal.h"

namespace {

constexpr int MAX_ARR = 16;
constexpr size_t ALIGN = 256;

struct HArr {
  const void *src = nullptr;   // host source (inputs)
  void *dst = nullptr;         // host destination (outputs); an output with dst == nullptr is not wanted
  size_t row = 0;              // bytes per item, or total bytes when `shared`
  bool shared = false;         // the same bytes for every chunk (key rows)
  bool direct = false;         // host memory is pinned: DMA straight from / to it
  bool temp = false;           // lives on the device only (an intermediate of a multi-stage chunk): no copy either way
  size_t dev_off = 0, pin_off = 0;
};

inline size_t up(size_t v) { return (v + ALIGN - 1) & ~(ALIGN - 1); }

// True when `p` points into memory HIP knows as pinned host memory (hipHostMalloc / hipHostRegister).
bool is_pinned(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

// memcpy on several threads once the block is large enough for the extra threads to pay for themselves.
void big_memcpy(void *dst, const void *src, size_t bytes) {
  constexpr size_t PER_THREAD = (size_t)4 << 20;
  unsigned hw = std::thread::hardware_concurrency();
  size_t nt = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), bytes / PER_THREAD);
  if (nt <= 1) { memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  const size_t part = ((bytes / nt) + 63) & ~(size_t)63;
  for (size_t t = 1; t < nt; t++) {
    const size_t o = t * part, n = o >= bytes ? 0 : std::min(part, bytes - o);
    if (n) th.emplace_back([=] { memcpy((char *)dst + o, (const char *)src + o, n); });
  }
  memcpy(dst, src, std::min(part, bytes));
  for (auto &t : th) t.join();
}

struct Pending { void *dst; const void *pin; size_t bytes; };

struct Pipeline {
  ntru_engine *eng;
  HArr arr[MAX_ARR];
  int n = 0;
  std::vector<Pending> pending[NTRU_HOST_SLOTS];
  hipStream_t saved_stream;
  GrowBuf *saved_scratch;

  explicit Pipeline(ntru_engine *e) : eng(e), saved_stream(e->stream), saved_scratch(e->cur_scratch) {}
  ~Pipeline() { eng->stream = saved_stream; eng->cur_scratch = saved_scratch; }

  int in(const void *p, size_t row, bool shared = false) {
    arr[n].src = p; arr[n].row = row; arr[n].shared = shared; arr[n].direct = !shared && is_pinned(p);
    return n++;
  }
  int out(void *p, size_t row) {
    arr[n].dst = p; arr[n].row = row; arr[n].direct = p && is_pinned(p);
    return n++;
  }
  int tmp(size_t row) {          // device-only rows of a chunk (what one stage hands the next)
    arr[n].row = row; arr[n].temp = true;
    return n++;
  }

  // Waits until the chunk that owns buffer set s has been downloaded, then hands its staged outputs to the caller's arrays.
  int drain(int s) {
    HostSlot &sl = eng->slot[s];
    if (!sl.busy) return NTRU_OK;
    HIP_TRY(hipEventSynchronize(sl.down_done));
    sl.busy = false;
    for (const Pending &p : pending[s]) big_memcpy(p.dst, p.pin, p.bytes);
    pending[s].clear();
    return NTRU_OK;
  }

  // launch(first item, items, device pointers in the order the arrays were declared) enqueues on eng->stream.
  template <class F>
  int run(int64_t B, int64_t C, F launch) {
    HIP_TRY(hipSetDevice(eng->device));
    if (C > B) C = B;
    if (C < 1) C = 1;
    size_t dev_bytes = 0, pin_bytes = 0;
    for (int i = 0; i < n; i++) {
      HArr &a = arr[i];
      const size_t bytes = a.shared ? a.row : a.row * (size_t)C;
      if (!a.src && !a.dst && !a.temp) continue;
      a.dev_off = dev_bytes; dev_bytes += up(bytes);
      if (!a.direct && !a.temp) { a.pin_off = pin_bytes; pin_bytes += up(bytes); }
    }
    for (hipStream_t *st : {&eng->st_up, &eng->st_comp, &eng->st_down})
      if (!*st) HIP_TRY(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    const int64_t nchunks = (B + C - 1) / C;
    // A single chunk (every call of the reference's own API) has nothing to overlap with: its three stages go onto ONE stream, in
    // order, without events between them.
    const bool single = nchunks == 1;
    const hipStream_t s_up = single ? eng->st_comp : eng->st_up, s_down = single ? eng->st_comp : eng->st_down;
    for (int s = 0; s < NTRU_HOST_SLOTS && s < nchunks; s++) {          // a single chunk touches one buffer set only
      HostSlot &sl = eng->slot[s];
      for (hipEvent_t *ev : {&sl.up_done, &sl.comp_done, &sl.down_done})
        if (!*ev) HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
      if (int rc = ntru_grow_dev(&sl.dev, dev_bytes)) return rc;
      if (int rc = ntru_grow_pinned(&sl.pinned, pin_bytes)) return rc;
    }
    int rc = NTRU_OK;
    int64_t k = 0;
    for (int64_t o = 0; o < B && rc == NTRU_OK; o += C, k++) {
      const int s = (int)(k % NTRU_HOST_SLOTS);
      const int64_t cnt = std::min(C, B - o);
      HostSlot &sl = eng->slot[s];
      if ((rc = drain(s))) break;                        // chunk k - 3 is out of this buffer set
      // ---- stage 1: upload (staging copies on this thread, DMA on the upload stream)
      void *dev[MAX_ARR];
      for (int i = 0; i < n && rc == NTRU_OK; i++) {
        HArr &a = arr[i];
        dev[i] = (a.src || a.dst || a.temp) ? (char *)sl.dev.p + a.dev_off : nullptr;
        if (!a.src) continue;
        const size_t bytes = a.shared ? a.row : a.row * (size_t)cnt;
        const char *from = (const char *)a.src + (a.shared ? 0 : a.row * (size_t)o);
        if (!a.direct) {
          char *pin = (char *)sl.pinned.p + a.pin_off;
          big_memcpy(pin, from, bytes);
          from = pin;
        }
        if (hipMemcpyAsync(dev[i], from, bytes, hipMemcpyHostToDevice, s_up) != hipSuccess)
          rc = ntru_fail(NTRU_ERR_HIP, "hipMemcpyAsync (host to device) failed");
      }
      if (!single && rc == NTRU_OK && hipEventRecord(sl.up_done, s_up) != hipSuccess) rc = ntru_fail(NTRU_ERR_HIP, "hipEventRecord failed");
      // ---- stage 2: compute, behind this chunk's upload
      if (!single && rc == NTRU_OK && hipStreamWaitEvent(eng->st_comp, sl.up_done, 0) != hipSuccess) rc = ntru_fail(NTRU_ERR_HIP, "hipStreamWaitEvent failed");
      eng->stream = eng->st_comp;
      eng->cur_scratch = &sl.scratch;
      if (rc == NTRU_OK) rc = launch(o, cnt, dev);
      if (!single && rc == NTRU_OK && hipEventRecord(sl.comp_done, eng->st_comp) != hipSuccess) rc = ntru_fail(NTRU_ERR_HIP, "hipEventRecord failed");
      // ---- stage 3: download, behind this chunk's kernels
      if (!single && rc == NTRU_OK && hipStreamWaitEvent(s_down, sl.comp_done, 0) != hipSuccess) rc = ntru_fail(NTRU_ERR_HIP, "hipStreamWaitEvent failed");
      for (int i = 0; i < n && rc == NTRU_OK; i++) {
        HArr &a = arr[i];
        if (!a.dst) continue;
        const size_t bytes = a.row * (size_t)cnt;
        char *to = (char *)a.dst + a.row * (size_t)o;
        if (!a.direct) {
          char *pin = (char *)sl.pinned.p + a.pin_off;
          pending[s].push_back({to, pin, bytes});
          to = pin;
        }
        if (hipMemcpyAsync(to, dev[i], bytes, hipMemcpyDeviceToHost, s_down) != hipSuccess)
          rc = ntru_fail(NTRU_ERR_HIP, "hipMemcpyAsync (device to host) failed");
      }
      // (recorded even after a failure: whatever was enqueued for this set must be waited for before the set is reused)
      if (hipEventRecord(sl.down_done, s_down) == hipSuccess) sl.busy = true;
      else if (rc == NTRU_OK) rc = ntru_fail(NTRU_ERR_HIP, "hipEventRecord failed");
    }
    // results of the chunks still in flight, oldest first; on failure still wait so nothing is left in flight
    const std::string err = rc ? std::string(ntru_last_error()) : std::string();
    for (int t = 0; t < NTRU_HOST_SLOTS; t++) {
      const int s = (int)((k + t) % NTRU_HOST_SLOTS);
      if (rc) {
        HostSlot &sl = eng->slot[s];
        if (sl.busy) { (void)hipStreamSynchronize(eng->st_up); (void)hipStreamSynchronize(eng->st_comp); (void)hipStreamSynchronize(eng->st_down); sl.busy = false; }
        pending[s].clear();
      } else rc = drain(s);
    }
    if (!err.empty()) ntru_fail(rc, err);
    return rc;
  }
};

// Items per chunk: large enough that a kernel launch fills the chip, small enough that a batch has several chunks in
// flight (a chunk of 2^15 N=821 round trips is ~0.4 GB over PCIe, ~7 ms; its kernels take ~0.15 ms).
int64_t chunk_items(int64_t B) {
  const int64_t big = 1 << 15;
  if (B >= 4 * big) return big;
  if (B >= 4 * 2048) return (B + 3) / 4;
  return B;
}

}  // namespace

extern "C" void *ntru_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    ntru_fail(NTRU_ERR_HIP, "hipHostMalloc failed");
    return nullptr;
  }
  return p;
}

extern "C" void ntru_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

#define CHECK_ENGINE()                                                      \
  if (!eng) return ntru_fail(NTRU_ERR_ARG, "engine is NULL");               \
  if (B < 0) return ntru_fail(NTRU_ERR_ARG, "negative batch size")

extern "C" int ntru_encrypt_batch(ntru_engine_t *eng, int N, int q, const uint16_t *h, const uint8_t *r,
                                  const uint8_t *m, int64_t B, uint16_t *e, uint16_t *quotE) {
  CHECK_ENGINE();
  if (int rc = ntru_encrypt_batch_dev(eng, N, q, nullptr, nullptr, nullptr, 0, nullptr, nullptr)) return rc;   // parameter checks
  if (B == 0) return NTRU_OK;
  if (!h || !r || !m || !e) return ntru_fail(NTRU_ERR_ARG, "ntru_encrypt_batch: NULL buffer");
  Pipeline P(eng);
  const int ih = P.in(h, (size_t)N * 2, true), ir = P.in(r, N), im = P.in(m, N), ie = P.out(e, (size_t)N * 2),
            iq = P.out(quotE, (size_t)N * 2);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_encrypt_batch_dev(eng, N, q, (const uint16_t *)d[ih], (const uint8_t *)d[ir], (const uint8_t *)d[im], n,
                                  (uint16_t *)d[ie], (uint16_t *)d[iq]);
  });
}

extern "C" int ntru_decrypt_batch(ntru_engine_t *eng, int N, int q, int p, const int8_t *f, const uint8_t *fp,
                                  const uint16_t *e, int64_t B, uint8_t *value, uint16_t *quot1, uint16_t *rem1,
                                  uint8_t *quot2) {
  CHECK_ENGINE();
  if (int rc = ntru_decrypt_batch_dev(eng, N, q, p, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr)) return rc;
  if (B == 0) return NTRU_OK;
  if (!f || !fp || !e || !value) return ntru_fail(NTRU_ERR_ARG, "ntru_decrypt_batch: NULL buffer");
  Pipeline P(eng);
  const int jf = P.in(f, N, true), jfp = P.in(fp, N, true), je = P.in(e, (size_t)N * 2), jv = P.out(value, N),
            jq1 = P.out(quot1, (size_t)N * 2), jr1 = P.out(rem1, (size_t)N * 2), jq2 = P.out(quot2, N);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_decrypt_batch_dev(eng, N, q, p, (const int8_t *)d[jf], (const uint8_t *)d[jfp], (const uint16_t *)d[je], n,
                                  (uint8_t *)d[jv], (uint16_t *)d[jq1], (uint16_t *)d[jr1], (uint8_t *)d[jq2]);
  });
}

extern "C" int ntru_polymul_split(ntru_engine_t *eng, int N, int mod, const uint16_t *a, const uint16_t *b,
                                  int64_t B, uint16_t *quot, uint16_t *rem) {
  CHECK_ENGINE();
  if (int rc = ntru_polymul_split_dev(eng, N, mod, nullptr, nullptr, 0, nullptr, nullptr)) return rc;
  if (B == 0) return NTRU_OK;
  if (!a || !b || !quot || !rem) return ntru_fail(NTRU_ERR_ARG, "ntru_polymul_split: NULL buffer");
  Pipeline P(eng);
  const size_t row = (size_t)N * 2;
  const int ia = P.in(a, row), ib = P.in(b, row), iq = P.out(quot, row), ir = P.out(rem, row);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_polymul_split_dev(eng, N, mod, (const uint16_t *)d[ia], (const uint16_t *)d[ib], n, (uint16_t *)d[iq],
                                  (uint16_t *)d[ir]);
  });
}

extern "C" int ntru_invert_key_batch(ntru_engine_t *eng, int N, int q, int p, const int8_t *f, int64_t B, uint16_t *fq,
                                     uint8_t *fp, uint8_t *flags) {
  CHECK_ENGINE();
  if (int rc = ntru_invert_key_batch_dev(eng, N, q, p, nullptr, 0, nullptr, nullptr, nullptr)) return rc;   // incl. p == 3
  if (B == 0) return NTRU_OK;
  if (!f || (!fq && !fp) || !flags) return ntru_fail(NTRU_ERR_ARG, "ntru_invert_key_batch: NULL buffer");
  Pipeline P(eng);
  const int jf = P.in(f, N), jfq = P.out(fq, (size_t)N * 2), jfp = P.out(fp, N), jfl = P.out(flags, 1);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_invert_key_batch_dev(eng, N, q, p, (const int8_t *)d[jf], n, (uint16_t *)d[jfq], (uint8_t *)d[jfp],
                                     (uint8_t *)d[jfl]);
  });
}

extern "C" int ntru_public_key_batch(ntru_engine_t *eng, int N, int q, int p, const uint16_t *fq, const int8_t *g,
                                     int64_t B, uint16_t *h) {
  CHECK_ENGINE();
  if (int rc = ntru_public_key_batch_dev(eng, N, q, p, nullptr, nullptr, 0, nullptr)) return rc;
  if (B == 0) return NTRU_OK;
  if (!fq || !g || !h) return ntru_fail(NTRU_ERR_ARG, "ntru_public_key_batch: NULL buffer");
  Pipeline P(eng);
  const int jfq = P.in(fq, (size_t)N * 2), jg = P.in(g, N), jh = P.out(h, (size_t)N * 2);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_public_key_batch_dev(eng, N, q, p, (const uint16_t *)d[jfq], (const int8_t *)d[jg], n, (uint16_t *)d[jh]);
  });
}

extern "C" int ntru_verify_keys_batch(ntru_engine_t *eng, int N, int q, int p, const int8_t *f, const int8_t *g,
                                      const uint16_t *fq, const uint8_t *fp, const uint16_t *h, int64_t B,
                                      uint16_t *quot_fq, uint16_t *rem_fq, uint8_t *quot_fp, uint8_t *rem_fp,
                                      uint16_t *quot_h, uint16_t *rem_h, uint8_t *flags) {
  CHECK_ENGINE();
  if (int rc = ntru_verify_keys_batch_dev(eng, N, q, p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                          nullptr, nullptr, nullptr, nullptr, nullptr)) return rc;
  if (B == 0) return NTRU_OK;
  if (!f || !g || !fq || !fp || !h || !quot_fq || !rem_fq || !quot_fp || !rem_fp || !quot_h || !rem_h || !flags)
    return ntru_fail(NTRU_ERR_ARG, "ntru_verify_keys_batch: NULL buffer");
  Pipeline P(eng);
  const size_t r8 = (size_t)N, r16 = (size_t)N * 2;
  const int jf = P.in(f, r8), jg = P.in(g, r8), jfq = P.in(fq, r16), jfp = P.in(fp, r8), jh = P.in(h, r16);
  const int o1 = P.out(quot_fq, r16), o2 = P.out(rem_fq, r16), o3 = P.out(quot_fp, r8), o4 = P.out(rem_fp, r8),
            o5 = P.out(quot_h, r16), o6 = P.out(rem_h, r16), ofl = P.out(flags, 1);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_verify_keys_batch_dev(eng, N, q, p, (const int8_t *)d[jf], (const int8_t *)d[jg], (const uint16_t *)d[jfq],
                                      (const uint8_t *)d[jfp], (const uint16_t *)d[jh], n, (uint16_t *)d[o1], (uint16_t *)d[o2],
                                      (uint8_t *)d[o3], (uint8_t *)d[o4], (uint16_t *)d[o5], (uint16_t *)d[o6], (uint8_t *)d[ofl]);
  });
}

extern "C" int ntru_split_by_I(ntru_engine_t *eng, int N, int mod, const uint16_t *a, int64_t B, uint16_t *quot,
                               uint16_t *rem) {
  CHECK_ENGINE();
  if (int rc = ntru_split_by_I_dev(eng, N, mod, nullptr, 0, nullptr, nullptr)) return rc;
  if (B == 0) return NTRU_OK;
  if (!a || !quot || !rem) return ntru_fail(NTRU_ERR_ARG, "ntru_split_by_I: NULL buffer");
  Pipeline P(eng);
  const size_t row = (size_t)N * 2;
  const int ia = P.in(a, 2 * row), iq = P.out(quot, row), ir = P.out(rem, row);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_split_by_I_dev(eng, N, mod, (const uint16_t *)d[ia], n, (uint16_t *)d[iq], (uint16_t *)d[ir]);
  });
}

extern "C" int ntru_add_batch(ntru_engine_t *eng, int N, int mod, const uint16_t *a, const uint16_t *b, int64_t B,
                              uint16_t *out) {
  CHECK_ENGINE();
  if (int rc = ntru_add_batch_dev(eng, N, mod, nullptr, nullptr, 0, nullptr)) return rc;
  if (B == 0) return NTRU_OK;
  if (!a || !b || !out) return ntru_fail(NTRU_ERR_ARG, "ntru_add_batch: NULL buffer");
  Pipeline P(eng);
  const size_t row = (size_t)N * 2;
  const int ia = P.in(a, row), ib = P.in(b, row), io = P.out(out, row);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_add_batch_dev(eng, N, mod, (const uint16_t *)d[ia], (const uint16_t *)d[ib], n, (uint16_t *)d[io]);
  });
}

extern "C" int ntru_sample_ternary(ntru_engine_t *eng, int N, int n1, int n2, int other, const uint32_t *key,
                                   uint64_t first_item, int64_t B, uint8_t *out) {
  if (!eng) return ntru_fail(NTRU_ERR_ARG, "engine is NULL");
  if (B > 0 && !out) return ntru_fail(NTRU_ERR_ARG, "ntru_sample_ternary: NULL buffer");
  if (B <= 0) return ntru_sample_ternary_dev(eng, N, n1, n2, other, key, first_item, B, nullptr);
  if (int rc = ntru_sample_ternary_dev(eng, N, n1, n2, other, key, first_item, 0, nullptr)) return rc;
  Pipeline P(eng);
  const int io = P.out(out, N);
  return P.run(B, chunk_items(B), [&](int64_t o, int64_t n, void **d) {
    return ntru_sample_ternary_dev(eng, N, n1, n2, other, key, first_item + (uint64_t)o, n, (uint8_t *)d[io]);
  });
}

extern "C" int ntru_pack_batch(ntru_engine_t *eng, int max_val, int data_len, const uint16_t *data, int64_t B,
                               uint64_t *out) {
  if (!eng) return ntru_fail(NTRU_ERR_ARG, "engine is NULL");
  int bits, per, al, os;
  if (int rc = ntru_pack_params(max_val, data_len, &bits, &per, &al, &os)) return rc;
  if (B <= 0) return B == 0 ? NTRU_OK : ntru_fail(NTRU_ERR_ARG, "negative batch size");
  if ((!data && data_len) || !out) return ntru_fail(NTRU_ERR_ARG, "ntru_pack_batch: NULL buffer");
  Pipeline P(eng);
  const int ii = P.in(data_len ? data : nullptr, (size_t)data_len * 2), io = P.out(out, (size_t)os * 32);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_pack_batch_dev(eng, max_val, data_len, (const uint16_t *)d[ii], n, (uint64_t *)d[io]);
  });
}

extern "C" int ntru_unpack_batch(ntru_engine_t *eng, int max_val, int packed_bits, const uint64_t *in, int packed_size,
                                 int64_t B, uint16_t *out) {
  if (!eng) return ntru_fail(NTRU_ERR_ARG, "engine is NULL");
  if (int rc = ntru_unpack_batch_dev(eng, max_val, packed_bits, nullptr, packed_size, 0, nullptr)) return rc;
  if (B < 0) return ntru_fail(NTRU_ERR_ARG, "negative batch size");
  if (B == 0 || packed_size == 0) return NTRU_OK;
  if (!in || !out) return ntru_fail(NTRU_ERR_ARG, "ntru_unpack_batch: NULL buffer");
  int bits = 0;
  while ((max_val >> bits) != 0) bits++;
  const int per = packed_bits / bits;
  Pipeline P(eng);
  const int ii = P.in(in, (size_t)packed_size * 32), io = P.out(out, (size_t)packed_size * per * 2);
  return P.run(B, chunk_items(B), [&](int64_t, int64_t n, void **d) {
    return ntru_unpack_batch_dev(eng, max_val, packed_bits, (const uint64_t *)d[ii], packed_size, n, (uint16_t *)d[io]);
  });
}

// ---- device-resident stages for a caller without HIP of its own (Node.js) -------------------------------------------------------
// ntru_pipeline_batch: sampler -> encryptBits -> decryptBits -> packOutput per chunk, the intermediates (r, e, value) staying in the
// slot's device arena; only m (and r when the caller supplies it) crosses PCIe upwards and only the outputs asked for come back
// (index.js:461-488, :87-140, :572-620 chained).  The chunks flow through the same three-stage pipeline (upload / compute / download
// streams, three buffer sets) as every host-pointer entry point, so the upload of chunk k+1 and the download of chunk k-1 overlap
// the kernels of chunk k.
extern "C" int ntru_pipeline_batch(ntru_engine_t *eng, int N, int q, int p, const uint16_t *h, const int8_t *f, const uint8_t *fp,
                                   const uint32_t *key, uint64_t first_item, int n1, int n2, const uint8_t *r, const uint8_t *m,
                                   int64_t B, uint8_t *r_out, uint16_t *e, uint8_t *value, uint64_t *packed) {
  CHECK_ENGINE();
  const bool decrypt = f != nullptr || fp != nullptr;
  if (decrypt && (!f || !fp)) return ntru_fail(NTRU_ERR_ARG, "ntru_pipeline_batch: the decrypt stage needs both f and fp");
  if (!decrypt && value) return ntru_fail(NTRU_ERR_ARG, "ntru_pipeline_batch: `value` needs the decrypt stage (f, fp)");
  if ((key != nullptr) == (r != nullptr)) return ntru_fail(NTRU_ERR_ARG, "ntru_pipeline_batch: give either a sampler key or r");
  if (!e && !value && !packed && !(r_out && key)) return ntru_fail(NTRU_ERR_ARG, "ntru_pipeline_batch: no output asked for");
  // parameter checks of every stage (B = 0 calls return after them)
  if (int rc = ntru_encrypt_batch_dev(eng, N, q, nullptr, nullptr, nullptr, 0, nullptr, nullptr)) return rc;
  if (decrypt) if (int rc = ntru_decrypt_batch_dev(eng, N, q, p, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr)) return rc;
  if (key) if (int rc = ntru_sample_ternary_dev(eng, N, n1, n2, p - 1, key, first_item, 0, nullptr)) return rc;
  int bits = 0, per = 0, al = 0, os = 0;
  const int pack_max = decrypt ? p - 1 : q - 1;                    // packOutput of the last stage's result
  if (packed) if (int rc = ntru_pack_params(pack_max, N, &bits, &per, &al, &os)) return rc;
  if (B == 0) return NTRU_OK;
  if (!h || !m) return ntru_fail(NTRU_ERR_ARG, "ntru_pipeline_batch: NULL buffer");
  Pipeline P(eng);
  const int ih = P.in(h, (size_t)N * 2, true), im = P.in(m, N);
  const int jf = decrypt ? P.in(f, N, true) : -1, jfp = decrypt ? P.in(fp, N, true) : -1;
  const int ir = r ? P.in(r, N) : (r_out ? P.out(r_out, N) : P.tmp(N));
  const int ie = e ? P.out(e, (size_t)N * 2) : P.tmp((size_t)N * 2);
  const int iv = !decrypt ? -1 : (value ? P.out(value, N) : P.tmp(N));
  const int ip = packed ? P.out(packed, (size_t)os * 32) : -1;
  return P.run(B, chunk_items(B), [&](int64_t o, int64_t n, void **d) {
    if (key) if (int rc = ntru_sample_ternary_dev(eng, N, n1, n2, p - 1, key, first_item + (uint64_t)o, n, (uint8_t *)d[ir])) return rc;
    if (!decrypt && packed) {   // packOutput of e: out of the encrypt kernel itself when e is not an output too and the row-image kernel applies
      if (!e && (eng->path == 0 || eng->path >= 4)) {       // the launcher itself says whether the fused kernel applies (NTRU_NOT_TAKEN:
        const int rc = ntru_launch_encrypt_pack_rowimage(eng, N, q, (const uint16_t *)d[ih], (const uint8_t *)d[ir], (const uint8_t *)d[im], n,
                                                         (uint64_t *)d[ip], os);     // e as the intermediate, below); no public error
        if (rc != NTRU_NOT_TAKEN) return rc;               // code doubles as control flow
      }
      return ntru_encrypt_pack_batch_dev(eng, N, q, (const uint16_t *)d[ih], (const uint8_t *)d[ir], (const uint8_t *)d[im], n,
                                         (uint16_t *)d[ie], (uint64_t *)d[ip]);
    }
    if (int rc = ntru_encrypt_batch_dev(eng, N, q, (const uint16_t *)d[ih], (const uint8_t *)d[ir], (const uint8_t *)d[im], n,
                                        (uint16_t *)d[ie], nullptr)) return rc;
    if (decrypt && packed)      // packOutput fused into the decrypt kernel's second epilogue where the matrix path applies (no k_pack launch)
      return ntru_decrypt_pack_batch_dev(eng, N, q, p, (const int8_t *)d[jf], (const uint8_t *)d[jfp], (const uint16_t *)d[ie], n,
                                         (uint8_t *)d[iv], (uint64_t *)d[ip]);
    if (decrypt)
      return ntru_decrypt_batch_dev(eng, N, q, p, (const int8_t *)d[jf], (const uint8_t *)d[jfp], (const uint16_t *)d[ie], n,
                                    (uint8_t *)d[iv], nullptr, nullptr, nullptr);
    return NTRU_OK;
  });
}

// Plain device buffers on the engine's device, with copies ordered on the engine's stream: what a binding needs to keep arrays
// on the GPU between *_dev calls (the addon hands them to JavaScript as opaque handles).
extern "C" int ntru_dev_alloc(ntru_engine_t *eng, size_t bytes, void **d_ptr) {
  if (!eng || !d_ptr) return ntru_fail(NTRU_ERR_ARG, "ntru_dev_alloc: NULL argument");
  *d_ptr = nullptr;
  HIP_TRY(hipSetDevice(eng->device));
  if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) { *d_ptr = nullptr; return ntru_fail(NTRU_ERR_HIP, "hipMalloc failed"); }
  return NTRU_OK;
}

extern "C" int ntru_dev_free(ntru_engine_t *eng, void *d_ptr) {
  if (!eng) return ntru_fail(NTRU_ERR_ARG, "engine is NULL");
  if (!d_ptr) return NTRU_OK;
  HIP_TRY(hipSetDevice(eng->device));
  HIP_TRY(hipFree(d_ptr));                  // waits for the device: nothing in flight still uses the buffer
  return NTRU_OK;
}

// Returns when `src` may be reused (pageable memory is copied before the call returns, pinned memory once the DMA is done); the data
// is in place for every later call on the engine's stream.
extern "C" int ntru_dev_upload(ntru_engine_t *eng, void *d_dst, const void *src, size_t bytes) {
  if (!eng || (bytes && (!d_dst || !src))) return ntru_fail(NTRU_ERR_ARG, "ntru_dev_upload: NULL argument");
  if (!bytes) return NTRU_OK;
  HIP_TRY(hipSetDevice(eng->device));
  HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, eng->stream));
  HIP_TRY(hipStreamSynchronize(eng->stream));
  return NTRU_OK;
}

// Waits for everything enqueued on the engine's stream, then returns with the bytes in `dst`.
extern "C" int ntru_dev_download(ntru_engine_t *eng, void *dst, const void *d_src, size_t bytes) {
  if (!eng || (bytes && (!dst || !d_src))) return ntru_fail(NTRU_ERR_ARG, "ntru_dev_download: NULL argument");
  HIP_TRY(hipSetDevice(eng->device));
  if (bytes) HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, eng->stream));
  HIP_TRY(hipStreamSynchronize(eng->stream));
  return NTRU_OK;
}

// ---- several devices in ONE process: contiguous shards, one host thread + engine (with its three stage streams) per device ------
// item b depends only on (key, m[b], r[b]) / (key, e[b]) / key[b], so a host batch is cut into contiguous
// slices [g B / G, (g + 1) B / G) and every slice runs the single-device pipeline above on its own thread; nothing is exchanged
// between devices.  (The benchmark's multi-GPU mode is one PROCESS per GPU instead -- bench.py under torchrun; this is for a
// host program, e.g. the Node.js addon, that owns all the GPUs of a node itself.)
struct ntru_multi {
  std::vector<ntru_engine_t *> eng;
};

extern "C" int ntru_multi_create(const int *device_ids, int n_dev, ntru_multi_t **out) {
  if (!out) return ntru_fail(NTRU_ERR_ARG, "ntru_multi_create: out is NULL");
  *out = nullptr;
  if (!device_ids || n_dev < 1 || n_dev > 64) return ntru_fail(NTRU_ERR_ARG, "ntru_multi_create: need 1 .. 64 device ids");
  ntru_multi *m = new ntru_multi;
  for (int i = 0; i < n_dev; i++) {
    ntru_engine_t *e = nullptr;
    if (int rc = ntru_engine_create(device_ids[i], &e)) {
      for (ntru_engine_t *x : m->eng) ntru_engine_destroy(x);
      delete m;
      return rc;
    }
    m->eng.push_back(e);
  }
  *out = m;
  return NTRU_OK;
}

extern "C" void ntru_multi_destroy(ntru_multi_t *m) {
  if (!m) return;
  for (ntru_engine_t *e : m->eng) ntru_engine_destroy(e);
  delete m;
}

extern "C" int ntru_multi_engines(const ntru_multi_t *m) { return m ? (int)m->eng.size() : 0; }

namespace {
// fn(engine, first item, items) on one thread per engine; the first failure (lowest shard) is what the caller sees
template <class F>
int for_each_shard(ntru_multi_t *m, int64_t B, F fn) {
  if (!m) return ntru_fail(NTRU_ERR_ARG, "multi-device engine is NULL");
  if (B < 0) return ntru_fail(NTRU_ERR_ARG, "negative batch size");
  const int G = (int)m->eng.size();
  std::vector<int> rc(G, NTRU_OK);
  std::vector<std::string> msg(G);
  std::vector<std::thread> th;
  auto shard = [&](int g) {
    const int64_t base = B / G, extra = B % G, lo = g * base + (g < extra ? g : extra), n = base + (g < extra ? 1 : 0);
    rc[g] = fn(m->eng[g], lo, n);
    if (rc[g]) msg[g] = ntru_last_error();               // the message is per thread: carry it over
  };
  for (int g = 1; g < G; g++) th.emplace_back(shard, g);
  shard(0);
  for (auto &t : th) t.join();
  for (int g = 0; g < G; g++)
    if (rc[g]) return ntru_fail(rc[g], "device shard " + std::to_string(g) + ": " + msg[g]);
  return NTRU_OK;
}
}  // namespace

extern "C" int ntru_multi_encrypt_batch(ntru_multi_t *m, int N, int q, const uint16_t *h, const uint8_t *r, const uint8_t *mm,
                                        int64_t B, uint16_t *e, uint16_t *quotE) {
  return for_each_shard(m, B, [&](ntru_engine_t *eng, int64_t lo, int64_t n) {
    const size_t o = (size_t)lo * N;
    return ntru_encrypt_batch(eng, N, q, h, r ? r + o : r, mm ? mm + o : mm, n, e ? e + o : e, quotE ? quotE + o : nullptr);
  });
}

extern "C" int ntru_multi_decrypt_batch(ntru_multi_t *m, int N, int q, int p, const int8_t *f, const uint8_t *fp,
                                        const uint16_t *e, int64_t B, uint8_t *value, uint16_t *quot1, uint16_t *rem1,
                                        uint8_t *quot2) {
  return for_each_shard(m, B, [&](ntru_engine_t *eng, int64_t lo, int64_t n) {
    const size_t o = (size_t)lo * N;
    return ntru_decrypt_batch(eng, N, q, p, f, fp, e ? e + o : e, n, value ? value + o : value, quot1 ? quot1 + o : nullptr,
                              rem1 ? rem1 + o : nullptr, quot2 ? quot2 + o : nullptr);
  });
}

extern "C" int ntru_multi_verify_keys_batch(ntru_multi_t *m, int N, int q, int p, const int8_t *f, const int8_t *g,
                                            const uint16_t *fq, const uint8_t *fp, const uint16_t *h, int64_t B,
                                            uint16_t *quot_fq, uint16_t *rem_fq, uint8_t *quot_fp, uint8_t *rem_fp,
                                            uint16_t *quot_h, uint16_t *rem_h, uint8_t *flags) {
  if (B > 0 && (!f || !g || !fq || !fp || !h || !quot_fq || !rem_fq || !quot_fp || !rem_fp || !quot_h || !rem_h || !flags))
    return ntru_fail(NTRU_ERR_ARG, "ntru_multi_verify_keys_batch: NULL buffer");
  return for_each_shard(m, B, [&](ntru_engine_t *eng, int64_t lo, int64_t n) {
    const size_t o = (size_t)lo * N;
    return ntru_verify_keys_batch(eng, N, q, p, f + o, g + o, fq + o, fp + o, h + o, n, quot_fq + o, rem_fq + o, quot_fp + o,
                                  rem_fp + o, quot_h + o, rem_h + o, flags + lo);
  });
}

extern "C" int ntru_multi_polymul_split(ntru_multi_t *m, int N, int mod, const uint16_t *a, const uint16_t *b, int64_t B,
                                        uint16_t *quot, uint16_t *rem) {
  if (B > 0 && (!a || !b || !quot || !rem)) return ntru_fail(NTRU_ERR_ARG, "ntru_multi_polymul_split: NULL buffer");
  return for_each_shard(m, B, [&](ntru_engine_t *eng, int64_t lo, int64_t n) {
    const size_t o = (size_t)lo * N;
    return ntru_polymul_split(eng, N, mod, a + o, b + o, n, quot + o, rem + o);
  });
}

extern "C" int ntru_multi_invert_key_batch(ntru_multi_t *m, int N, int q, int p, const int8_t *f, int64_t B, uint16_t *fq,
                                           uint8_t *fp, uint8_t *flags) {
  if (B > 0 && (!f || !flags)) return ntru_fail(NTRU_ERR_ARG, "ntru_multi_invert_key_batch: NULL buffer");
  return for_each_shard(m, B, [&](ntru_engine_t *eng, int64_t lo, int64_t n) {
    const size_t o = (size_t)lo * N;
    return ntru_invert_key_batch(eng, N, q, p, f + o, n, fq ? fq + o : nullptr, fp ? fp + o : nullptr, flags + lo);
  });
}

extern "C" int ntru_multi_public_key_batch(ntru_multi_t *m, int N, int q, int p, const uint16_t *fq, const int8_t *g, int64_t B,
                                           uint16_t *h) {
  if (B > 0 && (!fq || !g || !h)) return ntru_fail(NTRU_ERR_ARG, "ntru_multi_public_key_batch: NULL buffer");
  return for_each_shard(m, B, [&](ntru_engine_t *eng, int64_t lo, int64_t n) {
    const size_t o = (size_t)lo * N;
    return ntru_public_key_batch(eng, N, q, p, fq + o, g + o, n, h + o);
  });
}
