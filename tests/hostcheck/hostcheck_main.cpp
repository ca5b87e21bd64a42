// Host-side sanitizer driver: the real abi.hip + ntru_host.hip, compiled as plain C++ against the fake HIP runtime and the fake
// device (fake_hip.cpp, fake_device.cpp), run under AddressSanitizer + UBSan and under ThreadSanitizer.  What is exercised: engine
// life cycle, the three-stage chunk pipeline (pinned and pageable buffers, single chunk / many chunks / ragged last chunk / empty batch),
// optional outputs, the shared scratch buffer across two user streams, ntru_multi_* (one host thread per shard, unequal and empty
// shards), two engines driven from two threads, error paths.  Expected values come from the same formulas applied to the whole batch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "fake_formulas.h"
#include "ntru_engine.h"
#include "engine_internal.h"

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { g_fail++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)
#define OK(call) do { int rc_ = (call); CHECK(rc_ == 0, "%s -> %d (%s)", #call, rc_, ntru_last_error()); } while (0)

static thread_local uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

template <class T>
struct Buf {                 // pageable (std::vector) or pinned (ntru_host_alloc) host array
  std::vector<T> v;
  T *p = nullptr;
  size_t n = 0;
  bool pinned = false;
  Buf(size_t count, bool pin) : n(count), pinned(pin) {
    if (pin) p = (T *)ntru_host_alloc((count ? count : 1) * sizeof(T)); else { v.resize(count ? count : 1); p = v.data(); }
  }
  ~Buf() { if (pinned) ntru_host_free(p); }
  Buf(const Buf &) = delete;
};

static void round_trip(ntru_engine_t *eng, int N, int q, int64_t B, bool pin, bool witness) {
  const int p = 3;
  Buf<uint16_t> h(N, false), e(B * N, pin), quot(B * N, pin), e_want(B * N, false), quot_want(B * N, false);
  Buf<uint8_t> r(B * N, pin), m(B * N, pin);
  for (int i = 0; i < N; i++) h.p[i] = (uint16_t)(rnd() % q);
  for (int64_t i = 0; i < B * N; i++) { r.p[i] = (uint8_t)(rnd() % 3); m.p[i] = (uint8_t)(rnd() % 2); }
  OK(ntru_encrypt_batch(eng, N, q, h.p, r.p, m.p, B, e.p, witness ? quot.p : nullptr));
  fake_encrypt(N, q, h.p, r.p, m.p, B, e_want.p, quot_want.p);
  CHECK(memcmp(e.p, e_want.p, B * N * 2) == 0, "encrypt e, B=%ld pin=%d", (long)B, pin);
  if (witness) CHECK(memcmp(quot.p, quot_want.p, B * N * 2) == 0, "encrypt quot, B=%ld", (long)B);
  Buf<int8_t> f(N, false);
  Buf<uint8_t> fp(N, false), value(B * N, pin), q2(B * N, pin), value_want(B * N, false), q2_want(B * N, false);
  Buf<uint16_t> q1(B * N, pin), r1(B * N, pin), q1_want(B * N, false), r1_want(B * N, false);
  for (int i = 0; i < N; i++) { f.p[i] = (int8_t)((int)(rnd() % 3) - 1); fp.p[i] = (uint8_t)(rnd() % 3); }
  OK(ntru_decrypt_batch(eng, N, q, p, f.p, fp.p, e.p, B, value.p, witness ? q1.p : nullptr, witness ? r1.p : nullptr, witness ? q2.p : nullptr));
  fake_decrypt(N, q, p, f.p, fp.p, e.p, B, value_want.p, q1_want.p, r1_want.p, q2_want.p);
  CHECK(memcmp(value.p, value_want.p, B * N) == 0, "decrypt value, B=%ld pin=%d", (long)B, pin);
  if (witness) {
    CHECK(memcmp(q1.p, q1_want.p, B * N * 2) == 0 && memcmp(r1.p, r1_want.p, B * N * 2) == 0 && memcmp(q2.p, q2_want.p, B * N) == 0,
          "decrypt witness arrays, B=%ld", (long)B);
  }
}

static void per_item(ntru_engine_t *eng, int N, int q, int64_t B, bool pin) {
  const int p = 3;
  Buf<int8_t> f(B * N, pin), g(B * N, pin);
  Buf<uint16_t> fq(B * N, pin), h(B * N, pin), a(B * N, pin);
  Buf<uint8_t> fp(B * N, pin);
  for (int64_t i = 0; i < B * N; i++) { f.p[i] = (int8_t)((int)(rnd() % 3) - 1); g.p[i] = (int8_t)((int)(rnd() % 3) - 1); fq.p[i] = (uint16_t)(rnd() % q);
                                         h.p[i] = (uint16_t)(rnd() % q); a.p[i] = (uint16_t)(rnd() % q); fp.p[i] = (uint8_t)(rnd() % 3); }
  Buf<uint16_t> o1(B * N, pin), o2(B * N, pin), o5(B * N, pin), o6(B * N, pin), w1(B * N, false), w2(B * N, false), w5(B * N, false), w6(B * N, false);
  Buf<uint8_t> o3(B * N, pin), o4(B * N, pin), fl(B, pin), w3(B * N, false), w4(B * N, false), wf(B, false);
  OK(ntru_verify_keys_batch(eng, N, q, p, f.p, g.p, fq.p, fp.p, h.p, B, o1.p, o2.p, o3.p, o4.p, o5.p, o6.p, fl.p));
  fake_verify(N, q, p, f.p, g.p, fq.p, fp.p, h.p, B, w1.p, w2.p, w3.p, w4.p, w5.p, w6.p, wf.p);
  CHECK(!memcmp(o1.p, w1.p, B * N * 2) && !memcmp(o2.p, w2.p, B * N * 2) && !memcmp(o3.p, w3.p, B * N) && !memcmp(o4.p, w4.p, B * N) &&
        !memcmp(o5.p, w5.p, B * N * 2) && !memcmp(o6.p, w6.p, B * N * 2) && !memcmp(fl.p, wf.p, B), "verify_keys, B=%ld pin=%d", (long)B, pin);
  OK(ntru_polymul_split(eng, N, q, a.p, fq.p, B, o1.p, o2.p));
  fake_polymul(N, q, a.p, fq.p, B, w1.p, w2.p);
  CHECK(!memcmp(o1.p, w1.p, B * N * 2) && !memcmp(o2.p, w2.p, B * N * 2), "polymul_split, B=%ld", (long)B);
  OK(ntru_public_key_batch(eng, N, q, p, fq.p, g.p, B, o1.p));
  fake_public_key(N, q, p, fq.p, g.p, B, w1.p);
  CHECK(!memcmp(o1.p, w1.p, B * N * 2), "public_key, B=%ld", (long)B);
  OK(ntru_invert_key_batch(eng, N, q, p, f.p, B, o1.p, o3.p, fl.p));
  bool same = true;
  for (int64_t i = 0; i < B * N && same; i++) same = o1.p[i] == (uint16_t)((((f.p[i] + 3) & (q - 1)) * 5 + 1) & (q - 1)) && o3.p[i] == (uint8_t)((f.p[i] + 4) % p);
  for (int64_t b = 0; b < B && same; b++) same = fl.p[b] == (uint8_t)(f.p[b * N] & 1);
  CHECK(same, "invert_key (scratch buffer per slot), B=%ld", (long)B);
  OK(ntru_add_batch(eng, N, q, a.p, h.p, B, o1.p));
  same = true;
  for (int64_t i = 0; i < B * N && same; i++) same = o1.p[i] == (uint16_t)((a.p[i] + h.p[i]) % q);
  CHECK(same, "add_batch, B=%ld", (long)B);
  uint32_t key[8] = {9, 8, 7, 6, 5, 4, 3, 2};
  OK(ntru_sample_ternary(eng, N, 3, 3, 2, key, 1000, B, o3.p));
  same = true;
  for (int64_t b = 0; b < B && same; b++) for (int i = 0; i < N && same; i++) same = o3.p[b * N + i] == (uint8_t)((1000 + b) * 7 + i * 3 + 9 + 3 + 3 + 2);
  CHECK(same, "sample_ternary (item offsets across chunks), B=%ld", (long)B);
}

// ntru_pipeline_batch: device-only intermediates (r, e), optional outputs, several chunks
static void pipeline(ntru_engine_t *eng, int N, int q, int64_t B, bool pin) {
  const int p = 3;
  Buf<uint16_t> h(N, false), e(B * N, pin), e_want(B * N, false);
  Buf<int8_t> f(N, false);
  Buf<uint8_t> fp(N, false), m(B * N, pin), r(B * N, false), value(B * N, pin), value_want(B * N, false), r_back(B * N, pin);
  for (int i = 0; i < N; i++) { h.p[i] = (uint16_t)(rnd() % q); f.p[i] = (int8_t)((int)(rnd() % 3) - 1); fp.p[i] = (uint8_t)(rnd() % 3); }
  for (int64_t i = 0; i < B * N; i++) m.p[i] = (uint8_t)(rnd() % 2);
  uint32_t key[8] = {5, 4, 3, 2, 1, 0, 9, 8};
  for (int64_t b = 0; b < B; b++) for (int i = 0; i < N; i++) r.p[b * N + i] = (uint8_t)((77 + b) * 7 + i * 3 + 5 + 4 + 4 + 2);   // what the fake sampler draws
  fake_encrypt(N, q, h.p, r.p, m.p, B, e_want.p, nullptr);
  std::vector<uint16_t> d1(B * N ? B * N : 1), d2(B * N ? B * N : 1);
  std::vector<uint8_t> d3(B * N ? B * N : 1);
  fake_decrypt(N, q, p, f.p, fp.p, e_want.p, B, value_want.p, d1.data(), d2.data(), d3.data());
  OK(ntru_pipeline_batch(eng, N, q, p, h.p, f.p, fp.p, key, 77, 4, 4, nullptr, m.p, B, r_back.p, e.p, value.p, nullptr));
  CHECK(!memcmp(r_back.p, r.p, B * N) && !memcmp(e.p, e_want.p, B * N * 2) && !memcmp(value.p, value_want.p, B * N), "pipeline r / e / value, B=%ld pin=%d", (long)B, pin);
  memset(value.p, 0xEE, B * N);
  OK(ntru_pipeline_batch(eng, N, q, p, h.p, f.p, fp.p, key, 77, 4, 4, nullptr, m.p, B, nullptr, nullptr, value.p, nullptr));   // only m up, only value down
  CHECK(!memcmp(value.p, value_want.p, B * N), "lean pipeline, B=%ld", (long)B);
  OK(ntru_pipeline_batch(eng, N, q, p, h.p, nullptr, nullptr, nullptr, 0, 0, 0, r.p, m.p, B, nullptr, e.p, nullptr, nullptr));   // encrypt only, r given
  CHECK(!memcmp(e.p, e_want.p, B * N * 2), "encrypt-only pipeline, B=%ld", (long)B);
  CHECK(ntru_pipeline_batch(eng, N, q, p, h.p, f.p, fp.p, key, 0, 4, 4, r.p, m.p, B, nullptr, e.p, nullptr, nullptr) != 0, "key AND r must be refused");
  CHECK(ntru_pipeline_batch(eng, N, q, p, h.p, nullptr, nullptr, key, 0, 4, 4, nullptr, m.p, B, nullptr, nullptr, value.p, nullptr) != 0, "value without f, fp must be refused");
}

static void two_streams_share_the_scratch(ntru_engine_t *eng, int N, int q) {
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, 0); hipStreamCreateWithFlags(&sb, 0);
  const int64_t B = 300;
  std::vector<int8_t> f1(B * N), f2(B * N);
  std::vector<uint16_t> o1(B * N), o2(B * N);
  std::vector<uint8_t> p1(B * N), p2(B * N), l1(B), l2(B);
  for (auto &x : f1) x = (int8_t)((int)(rnd() % 3) - 1);
  for (auto &x : f2) x = (int8_t)((int)(rnd() % 3) - 1);
  OK(ntru_engine_set_stream(eng, sa));
  OK(ntru_invert_key_batch_dev(eng, N, q, 3, f1.data(), B, o1.data(), p1.data(), l1.data()));   // ("device" memory is the heap here)
  OK(ntru_engine_set_stream(eng, sb));
  OK(ntru_invert_key_batch_dev(eng, N, q, 3, f2.data(), B, o2.data(), p2.data(), l2.data()));   // must wait for stream a's use of the scratch
  hipStreamSynchronize(sa); hipStreamSynchronize(sb);
  bool same = true;
  for (int64_t i = 0; i < B * N && same; i++)
    same = o1[i] == (uint16_t)((((f1[i] + 3) & (q - 1)) * 5 + 1) & (q - 1)) && o2[i] == (uint16_t)((((f2[i] + 3) & (q - 1)) * 5 + 1) & (q - 1));
  CHECK(same, "two *_dev calls on two streams sharing the engine's scratch buffer");
  OK(ntru_engine_set_stream(eng, nullptr));
  hipStreamDestroy(sa); hipStreamDestroy(sb);
}

static void multi(int N, int q) {
  const int ids[3] = {0, 1, 2};
  ntru_multi_t *mu = nullptr;
  OK(ntru_multi_create(ids, 3, &mu));
  CHECK(ntru_multi_engines(mu) == 3, "three engines");
  for (int64_t B : {(int64_t)2, (int64_t)1000, (int64_t)70001}) {       // 2 items on 3 engines: an empty shard
    std::vector<uint16_t> h(N), e(B * N), quot(B * N), ew(B * N), qw(B * N);
    std::vector<uint8_t> r(B * N), m(B * N);
    for (auto &x : h) x = (uint16_t)(rnd() % q);
    for (int64_t i = 0; i < B * N; i++) { r[i] = (uint8_t)(rnd() % 3); m[i] = (uint8_t)(rnd() % 2); }
    OK(ntru_multi_encrypt_batch(mu, N, q, h.data(), r.data(), m.data(), B, e.data(), quot.data()));
    fake_encrypt(N, q, h.data(), r.data(), m.data(), B, ew.data(), qw.data());
    CHECK(e == ew && quot == qw, "multi encrypt, B=%ld", (long)B);
    std::vector<int8_t> f(N);
    std::vector<uint8_t> fp(N), v(B * N), vw(B * N);
    for (int i = 0; i < N; i++) { f[i] = (int8_t)((int)(rnd() % 3) - 1); fp[i] = (uint8_t)(rnd() % 3); }
    OK(ntru_multi_decrypt_batch(mu, N, q, 3, f.data(), fp.data(), e.data(), B, v.data(), nullptr, nullptr, nullptr));
    std::vector<uint16_t> d1(B * N), d2(B * N);
    std::vector<uint8_t> d3(B * N);
    fake_decrypt(N, q, 3, f.data(), fp.data(), e.data(), B, vw.data(), d1.data(), d2.data(), d3.data());
    CHECK(v == vw, "multi decrypt (value only), B=%ld", (long)B);
  }
  ntru_multi_destroy(mu);
}

int main() {
  CHECK(ntru_engine_device_count() == 4, "fake device count");
  ntru_engine_t *eng = nullptr;
  OK(ntru_engine_create(0, &eng));
  ntru_engine_t *e2 = nullptr;
  CHECK(ntru_engine_create(9, &e2) != 0 && e2 == nullptr && strlen(ntru_last_error()) > 0, "device id out of range must fail with a message");
  OK(ntru_engine_create(1, &e2));
  CHECK(ntru_engine_set_kernel_path(eng, 77) != 0, "bad kernel path must be refused");
  {   // the occupancy query is corrected for the 1280-byte pieces LDS is handed out in (profiles/r03_wg_residency.txt)
    struct { int threads; size_t lds; int want; } cases[] = {{64, 13312, 11}, {64, 12800, 12}, {64, 12801, 11}, {64, 10241, 14},
                                                             {256, 81920, 2}, {256, 81921, 1}, {256, 0, 2}, {64, 5121, 25}};
    for (auto &c : cases) {
      int n = 0;
      OK(ntru_blocks_per_cu(eng, (const void *)&cases, c.threads, c.lds, &n));
      CHECK(n == c.want, "workgroups per CU after the LDS-piece correction");
    }
  }
  const int N = 61, q = 2048;
  for (bool pin : {false, true})
    for (int64_t B : {(int64_t)0, (int64_t)1, (int64_t)1000, (int64_t)(1 << 15) + 7, (int64_t)3 * (1 << 15) + 5}) {
      round_trip(eng, N, q, B, pin, true);
#ifndef HOSTCHECK_QUICK                                      // (the ThreadSanitizer build runs ~20x slower: fewer repetitions, same paths)
      round_trip(eng, N, q, B, pin, false);
#endif
    }
  for (bool pin : {false, true})
#ifdef HOSTCHECK_QUICK
    for (int64_t B : {(int64_t)1, (int64_t)(1 << 15) + 9}) per_item(eng, N, q, B, pin);
#else
    for (int64_t B : {(int64_t)1, (int64_t)(1 << 15) + 9, (int64_t)2 * (1 << 15) + 1}) per_item(eng, N, q, B, pin);
#endif
  round_trip(eng, N, q, 1000, false, false);
  for (bool pin : {false, true})
    for (int64_t B : {(int64_t)1, (int64_t)9001, (int64_t)4 * (1 << 15) + 3}) pipeline(eng, N, q, B, pin);
  {   // plain device buffers
    void *d = nullptr;
    OK(ntru_dev_alloc(eng, 4096, &d));
    std::vector<uint8_t> a(4096), b(4096);
    for (auto &x : a) x = (uint8_t)rnd();
    OK(ntru_dev_upload(eng, d, a.data(), a.size()));
    OK(ntru_dev_download(eng, b.data(), d, b.size()));
    CHECK(a == b, "dev_upload / dev_download");
    OK(ntru_dev_free(eng, d));
  }
  two_streams_share_the_scratch(eng, N, q);
  {   // two engines from two host threads at once (thread-local error strings, nothing shared between engines)
    std::thread t1([&] { round_trip(eng, N, q, (1 << 15) + 11, false, true); });
    std::thread t2([&] { round_trip(e2, N, q, (1 << 15) + 13, true, true); });
    t1.join(); t2.join();
  }
  multi(N, q);
  CHECK(ntru_encrypt_batch(eng, N, 1000, nullptr, nullptr, nullptr, 4, nullptr, nullptr) != 0, "q that is no power of two must be refused");
  ntru_engine_destroy(e2);
  ntru_engine_destroy(eng);
  if (g_fail) { fprintf(stderr, "hostcheck: %d failure(s)\n", g_fail); return 1; }
  printf("hostcheck ok\n");
  return 0;
}
