// Test doubles of the kernel launchers (ntru_launch_*) and of the *_dev entry points that live next to their kernels: each enqueues,
// on the engine's current stream, a closure that computes a cheap deterministic function of ALL its inputs into ALL its outputs --
// enough for the driver to tell whether the host pipeline (chunking, three buffer sets, pinned / pageable staging, shards on several
// threads) moved every byte to the right place.  No NTRU arithmetic here: the product kernels are tested on the GPU.
#include <hip/hip_runtime.h>

#include <cstring>

#include "engine_internal.h"
#include "fake_formulas.h"

int ntru_launch_encrypt_matrix(ntru_engine *eng, int N, int q, int ld, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m,
                               int64_t B, uint16_t *d_e, uint16_t *d_quotE) {
  if (ld != N) return NTRU_NOT_TAKEN;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "fake_encrypt");
  fake_enqueue(eng->stream, [=] { fake_encrypt(N, q, d_h, d_r, d_m, B, d_e, d_quotE); });
  return NTRU_OK;
}
int ntru_launch_decrypt_pack_matrix(ntru_engine *, int, int, int, const int8_t *, const uint8_t *, const uint16_t *, int64_t, uint8_t *, uint64_t *, int) {
  return NTRU_NOT_TAKEN;      // the host pipeline then runs decrypt + pack as two (fake) launches
}
int ntru_launch_encrypt_rowimage(ntru_engine *, int, int, int, const uint16_t *, const uint8_t *, const uint8_t *, int64_t, uint16_t *, uint16_t *) {
  return NTRU_NOT_TAKEN;
}
int ntru_launch_encrypt_pack_rowimage(ntru_engine *, int, int, const uint16_t *, const uint8_t *, const uint8_t *, int64_t, uint64_t *, int) {
  return NTRU_NOT_TAKEN;      // the host pipeline then runs encrypt + pack as two (fake) launches
}
int ntru_launch_encrypt_valu(ntru_engine *, int, int, const uint16_t *, const uint8_t *, const uint8_t *, int64_t, uint16_t *, uint16_t *) {
  return ntru_fail(NTRU_ERR_UNSUPPORTED, "fake device: matrix launcher only");
}
int ntru_launch_decrypt_matrix(ntru_engine *eng, int N, int q, int p, int ld, const int8_t *d_f, const uint8_t *d_fp, const uint16_t *d_e,
                               int64_t B, uint8_t *d_value, uint16_t *d_quot1, uint16_t *d_rem1, uint8_t *d_quot2) {
  if (ld != N) return NTRU_NOT_TAKEN;
  fake_enqueue(eng->stream, [=] { fake_decrypt(N, q, p, d_f, d_fp, d_e, B, d_value, d_quot1, d_rem1, d_quot2); });
  return NTRU_OK;
}
int ntru_launch_decrypt_valu(ntru_engine *, int, int, int, const int8_t *, const uint8_t *, const uint16_t *, int64_t, uint8_t *, uint16_t *,
                             uint16_t *, uint8_t *) {
  return ntru_fail(NTRU_ERR_UNSUPPORTED, "fake device: matrix launcher only");
}
bool ntru_newton_round_matrix_applies(const ntru_engine *, int, int, int) { return false; }
int ntru_launch_newton_round_matrix(ntru_engine *, int, int, int, const int8_t *, uint16_t *, long) { return NTRU_NOT_TAKEN; }
int ntru_launch_polymul_matrix(ntru_engine *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b, int64_t B, uint16_t *d_quot,
                               uint16_t *d_rem) {
  fake_enqueue(eng->stream, [=] { fake_polymul(N, mod, d_a, d_b, B, d_quot, d_rem); });
  return NTRU_OK;
}
int ntru_launch_polymul_valu(ntru_engine *, int, int, const uint16_t *, const uint16_t *, int64_t, uint16_t *, uint16_t *) { return NTRU_ERR_UNSUPPORTED; }
bool ntru_product_tern_matrix_applies(const ntru_engine *, int, int) { return true; }
int ntru_launch_product_tern_matrix(ntru_engine *eng, int N, int q, uint32_t mul, const uint16_t *d_a, const int8_t *d_s, long B,
                                    uint16_t *d_rem) {
  fake_enqueue(eng->stream, [=] { fake_public_key(N, q, (int)mul, d_a, d_s, B, d_rem); });
  return NTRU_OK;
}
int ntru_launch_public_key_valu(ntru_engine *, int, int, int, const uint16_t *, const int8_t *, int64_t, uint16_t *) { return NTRU_ERR_UNSUPPORTED; }
int ntru_launch_verify_keys_matrix(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const int8_t *d_g, const uint16_t *d_fq,
                                   const uint8_t *d_fp, const uint16_t *d_h, int64_t B, uint16_t *d_quot_fq, uint16_t *d_rem_fq,
                                   uint8_t *d_quot_fp, uint8_t *d_rem_fp, uint16_t *d_quot_h, uint16_t *d_rem_h, uint8_t *d_flags) {
  fake_enqueue(eng->stream, [=] { fake_verify(N, q, p, d_f, d_g, d_fq, d_fp, d_h, B, d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp, d_quot_h, d_rem_h, d_flags); });
  return NTRU_OK;
}
int ntru_launch_verify_keys_valu(ntru_engine *, int, int, int, const int8_t *, const int8_t *, const uint16_t *, const uint8_t *,
                                 const uint16_t *, int64_t, uint16_t *, uint16_t *, uint8_t *, uint8_t *, uint16_t *, uint16_t *, uint8_t *) {
  return NTRU_ERR_UNSUPPORTED;
}

// ---- the entry points of keygen_sampler_pack.hip -------------------------------------------------------------------------------
extern "C" int ntru_split_by_I_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, int64_t B, uint16_t *d_quot, uint16_t *d_rem) {
  if (B == 0) return NTRU_OK;
  fake_enqueue(eng->stream, [=] { for (int64_t i = 0; i < B * N; i++) { d_quot[i] = (uint16_t)((d_a[2 * i] + 1) % mod); d_rem[i] = (uint16_t)((d_a[2 * i + 1] + 2) % mod); } });
  return NTRU_OK;
}
extern "C" int ntru_add_batch_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b, int64_t B, uint16_t *d_out) {
  if (B == 0) return NTRU_OK;
  fake_enqueue(eng->stream, [=] { for (int64_t i = 0; i < B * N; i++) d_out[i] = (uint16_t)((d_a[i] + d_b[i]) % mod); });
  return NTRU_OK;
}
extern "C" int ntru_invert_key_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f, int64_t B, uint16_t *d_fq,
                                         uint8_t *d_fp, uint8_t *d_flags) {
  if (B == 0) return NTRU_OK;
  char *sc = nullptr;                                           // exercises the shared scratch buffer like the real call
  if (int rc = ntru_scratch_acquire(eng, (size_t)B * N * 2, &sc)) return rc;
  uint16_t *tmp = (uint16_t *)sc;
  fake_enqueue(eng->stream, [=] {
    for (int64_t i = 0; i < B * N; i++) tmp[i] = (uint16_t)((d_f[i] + 3) & (q - 1));
    for (int64_t i = 0; i < B * N; i++) { if (d_fq) d_fq[i] = (uint16_t)((tmp[i] * 5 + 1) & (q - 1)); if (d_fp) d_fp[i] = (uint8_t)((d_f[i] + 4) % p); }
    for (int64_t b = 0; b < B; b++) d_flags[b] = (uint8_t)(d_f[b * N] & 1);
  });
  return ntru_scratch_release(eng);
}
extern "C" int ntru_sample_ternary_dev(ntru_engine_t *eng, int N, int n1, int n2, int other, const uint32_t *key, uint64_t first_item,
                                       int64_t B, uint8_t *d_out) {
  if (B <= 0) return NTRU_OK;
  const uint32_t k0 = key[0];
  fake_enqueue(eng->stream, [=] { for (int64_t b = 0; b < B; b++) for (int i = 0; i < N; i++) d_out[b * N + i] = (uint8_t)((first_item + b) * 7 + i * 3 + k0 + n1 + n2 + other); });
  return NTRU_OK;
}
extern "C" int ntru_pack_params(int max_val, int data_len, int *bits, int *per_output, int *arr_len, int *output_size) {
  int b = 0;
  while ((max_val >> b) != 0) b++;
  const int n = 252 / b;
  int al = ((data_len + n - 1) / n) * n;
  if (al < 3 * n) al = 3 * n;
  *bits = b; *per_output = n; *arr_len = al; *output_size = al / n < 3 ? 3 : al / n;
  return NTRU_OK;
}
extern "C" int ntru_pack_batch_dev(ntru_engine_t *eng, int max_val, int data_len, const uint16_t *d_data, int64_t B, uint64_t *d_out) {
  int bits, per, al, os;
  ntru_pack_params(max_val, data_len, &bits, &per, &al, &os);
  if (B == 0) return NTRU_OK;
  fake_enqueue(eng->stream, [=] { for (int64_t b = 0; b < B; b++) for (int j = 0; j < os * 4; j++) d_out[b * os * 4 + j] = (uint64_t)d_data[b * data_len + j % data_len] * 0x10001ull + j; });
  return NTRU_OK;
}
extern "C" int ntru_pack_bytes_batch_dev(ntru_engine_t *eng, int max_val, int data_len, const uint8_t *d_data, int64_t B, uint64_t *d_out) {
  int bits, per, al, os;
  ntru_pack_params(max_val, data_len, &bits, &per, &al, &os);
  if (B == 0) return NTRU_OK;
  fake_enqueue(eng->stream, [=] { for (int64_t b = 0; b < B; b++) for (int j = 0; j < os * 4; j++) d_out[b * os * 4 + j] = (uint64_t)d_data[b * data_len + j % data_len] * 0x101ull + j; });
  return NTRU_OK;
}
extern "C" int ntru_unpack_batch_dev(ntru_engine_t *eng, int max_val, int packed_bits, const uint64_t *d_in, int packed_size, int64_t B,
                                     uint16_t *d_out) {
  int bits = 0;
  while ((max_val >> bits) != 0) bits++;
  const int per = packed_bits / bits;
  if (B == 0 || packed_size == 0) return NTRU_OK;
  fake_enqueue(eng->stream, [=] { for (int64_t b = 0; b < B; b++) for (int j = 0; j < packed_size * per; j++) d_out[b * packed_size * per + j] = (uint16_t)(d_in[(b * packed_size + j / per) * 4] + j); });
  return NTRU_OK;
}
