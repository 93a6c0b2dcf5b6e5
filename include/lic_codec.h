/* Host-side entropy coder of the MI355X hot path.  The reference has NO entropy
 * coder or bitstream (it only estimates bits, RateDistortionLoss.py:13-18); this is the
 * piece a deployment needs next to the device kernels, and the check that estimated bpp ~ coded bpp.
 *
 * A carry-propagating 32-bit range coder (byte-wise renormalisation, 16-bit frequencies) driven by
 * the cumulative tables the device builds (lic_factorized_cdf_tables / lic_gmm_cdf_tables in lic.h):
 * table t has S symbols and S+1 uint32 entries, cum[0] = 0 < cum[1] < ... < cum[S] = 65536.
 * Symbol n is the integer  idx[n] = value - window_lo  of its table; 0 < idx < S-1 is coded
 * directly, idx <= 0 / idx >= S-1 code the edge symbol followed by an Elias-gamma escape of the
 * excess in equiprobable bits, so any integer is representable.  Serial by nature: runs on the host
 * CPU, one stream per latent tensor (as the north star states).  Plain C ABI, no GPU dependency. */
#ifndef LIC_CODEC_H
#define LIC_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum lic_codec_status {
  LIC_CODEC_OK = 0,
  LIC_CODEC_ERR_INVALID = -1,  /* null pointer, bad sizes, malformed table */
  LIC_CODEC_ERR_OVERFLOW = -2, /* output buffer too small */
  LIC_CODEC_ERR_CORRUPT = -3   /* bitstream ended early or decoded an impossible escape */
};

/* upper bound of the encoded size of n symbols (escapes included for |excess| < 2^31) */
size_t lic_rc_bound(int64_t n);

/* tables: [T][S+1]; table_of[n] selects the table of symbol n (NULL: table n, i.e. T == n).
 * idx[n]: value - window_lo.  Writes *nbytes <= cap bytes to out. */
int lic_rc_encode(const uint32_t* tables, const int32_t* table_of, int32_t S, const int32_t* idx, int64_t n,
                  uint8_t* out, size_t cap, size_t* nbytes);
/* inverse: idx_out[n] for the same tables */
int lic_rc_decode(const uint8_t* in, size_t nbytes, const uint32_t* tables, const int32_t* table_of, int32_t S,
                  int64_t n, int32_t* idx_out);

/* Streaming decoder: the tables of later symbols may depend on symbols already decoded (the
 * masked-conv context model, decoded wavefront by wavefront: codec.ContextCodec).  `in` must stay valid until
 * lic_rc_decoder_free. */
typedef struct lic_rc_decoder lic_rc_decoder;
lic_rc_decoder* lic_rc_decoder_new(const uint8_t* in, size_t nbytes);
int lic_rc_decoder_next(lic_rc_decoder* dec, const uint32_t* tables, const int32_t* table_of, int32_t S, int64_t n,
                        int32_t* idx_out);
void lic_rc_decoder_free(lic_rc_decoder* dec);

/* -sum log2(freq/65536) of the coded symbols (escape bits included): the ideal size of the stream
 * for these tables, to compare with 8 * nbytes */
double lic_rc_ideal_bits(const uint32_t* tables, const int32_t* table_of, int32_t S, const int32_t* idx,
                         int64_t n);

int lic_codec_version(void);

#ifdef __cplusplus
}
#endif
#endif
