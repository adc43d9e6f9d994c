/*
 * dxtlt_oracle.h -- CPU oracle for the BCn block transform hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * scalar ("portable32" / "generic") loops.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library
 * (libdxtlt_gfx950.so) never links, loads or calls anything in oracle/.
 *
 * Parity pin status: the reference (Rust, /root/reference) cannot be built in
 * this image (no cargo/rustc) and holds NO golden vector of transformed bytes
 * for any BC1/BC2/BC3 mode.  What it does hold, and what this oracle is checked
 * against in tests/test_oracle.py:
 *   - the three test-data generator known-answer vectors
 *     (bc1 test_prelude.rs:107-119, bc2 :586-606, bc3 :1058-1078),
 *   - the split_565_color_endpoints 3-pair vector (common .../tests.rs:140-152),
 *   - the round-trip-for-every-n harness (bc1 test_prelude.rs:154-317 and twins),
 *   - the YCoCg-R 13/16-colour round-trip sets (decorrelate.rs:413-446, avx2.rs:194-266),
 *   - the real-texture round trips on assets/tests/r2-256-bc{1,2,3}.dds.
 * In the strict sense the transformed bytes are PARITY UNPINNED (no reference-produced
 * vector exists and the reference cannot be run here); everything the reference does pin is checked.
 * Forward-byte parity therefore rests on the code-defined layout plus an
 * independently written numpy restatement (oracle/oracle_np.py) agreeing with
 * this file on every case; see DESIGN.md "Oracle".  The road to a real pin is one
 * command on a machine with cargo: tests/golden/reference_kit/run.sh writes the
 * reference's own output bytes, and tests/test_reference_vectors.py then holds
 * this oracle (and the HIP path) to them.
 *
 * All reference paths below are relative to /root/reference/src/core/.
 */
#ifndef DXTLT_ORACLE_H
#define DXTLT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* YCoCgVariant, core numbering:
 * dxt-lossless-transform-common/src/color_565/decorrelate.rs:72-84 */
enum {
    ORACLE_YCOCG_NONE = 0,
    ORACLE_YCOCG_VAR1 = 1,
    ORACLE_YCOCG_VAR2 = 2,
    ORACLE_YCOCG_VAR3 = 3
};

/* BcNValidationError (safe wrappers), e.g.
 * dxt-lossless-transform-bc1/src/transform/safe/transform_with_settings.rs:18-31 */
enum {
    ORACLE_OK = 0,
    ORACLE_INVALID_LENGTH = 1,
    ORACLE_OUTPUT_TOO_SMALL = 2
};

/* Color565::{decorrelate,recorrelate}_ycocg_r(variant), decorrelate.rs:364,391 */
uint16_t oracle_decorrelate_565(uint16_t v, int variant);
uint16_t oracle_recorrelate_565(uint16_t v, int variant);

/* transform_bcN_with_settings / untransform_bcN_with_settings (unsafe ptr API).
 * len is in bytes and must be a multiple of the block size. */
void oracle_transform_bc1(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_colour);
void oracle_untransform_bc1(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_colour);
void oracle_transform_bc2(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_colour);
void oracle_untransform_bc2(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_colour);
void oracle_transform_bc3(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_alpha,
                          int split_colour);
void oracle_untransform_bc3(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_alpha,
                            int split_colour);

/* *_safe wrappers: length / size validation then the call above. */
int oracle_transform_bc1_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                              int split_colour);
int oracle_untransform_bc1_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                                int split_colour);
int oracle_transform_bc2_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                              int split_colour);
int oracle_untransform_bc2_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                                int split_colour);
int oracle_transform_bc3_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                              int split_alpha, int split_colour);
int oracle_untransform_bc3_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                                int split_alpha, int split_colour);

/* The reference's deterministic test-data generators (test_prelude.rs). */
void oracle_generate_bc1_test_data(size_t num_blocks, uint8_t *out);
void oracle_generate_bc2_test_data(size_t num_blocks, uint8_t *out);
void oracle_generate_bc3_test_data(size_t num_blocks, uint8_t *out);

/* split_color_endpoints reference implementation
 * (common/src/transforms/split_565_color_endpoints/mod.rs:110): pairs -> all c0 then all c1. */
void oracle_split_565_color_endpoints(const uint8_t *in, uint8_t *out, size_t len_bytes);

/* Synthetic workload generator shared with the GPU fill kernel:
 * qword i of the buffer = splitmix64_mix(seed + (first_qword + i + 1) * GOLDEN), little-endian. */
void oracle_fill_splitmix64(uint8_t *out, size_t len_bytes, uint64_t seed, uint64_t first_qword);

/* 64-bit wrapping sum of the buffer read as little-endian u64 words (tail bytes zero-extended):
 * a cheap order-insensitive checksum computable on both sides. */
uint64_t oracle_sum_u64(const uint8_t *data, size_t len_bytes);

/* Multi-threaded (pthread) range split of the scalar loops above; used only by bench.py's
 * cpu_baseline leg.  kind: 1/2/3 = BC1/BC2/BC3; inverse != 0 runs the untransform. */
void oracle_run_mt(int kind, int inverse, const uint8_t *in, uint8_t *out, size_t len, int variant,
                   int split_alpha, int split_colour, int threads);

/* AVX2 port of the reference's SIMD strategy, BC1 default settings only (Variant1 + split colours), with a scalar
 * tail; equals oracle_{un,}transform_bc1(..., VAR1, 1) byte for byte.  cpu_baseline leg only.  (dxtlt_oracle_avx2.c) */
int oracle_simd_available(void);
/* 0 = scalar, 2 = AVX2, 5 = AVX-512BW: the widest level the CPU has (and the cap allows); set_cap returns the level now in effect */
int oracle_simd_level(void);
int oracle_simd_set_cap(int cap);
void oracle_bc1_default_simd_range(int inverse, const uint8_t *in, uint8_t *out, size_t n_total, size_t first,
                                   size_t count);
void oracle_bc1_default_simd_mt(int inverse, const uint8_t *in, uint8_t *out, size_t len, int threads);
/* AVX2 ports (scalar tail; scalar everywhere when the CPU lacks AVX2) of two more of the reference's vectorised paths:
 * kind 2 = BC2 default settings {Variant1, split colours}, kind 3 = BC3 "standard" {None, no splits}.  Equal to the
 * scalar oracle byte for byte (tests/test_oracle.py); cpu_baseline leg only.  (dxtlt_oracle_avx2.c) */
void oracle_bc23_simd_range(int kind, int inverse, const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count);
void oracle_bc23_simd_mt(int kind, int inverse, const uint8_t *in, uint8_t *out, size_t len, int threads);
/* one block range through the scalar loops */
void oracle_transform_range(int kind, int inverse, const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                            int variant, int split_alpha, int split_colour);

/* BC7 granule-sorted field split, version 2 -- a format defined by this build (docs/BC7_FORMAT.md); the reference has
 * no BC7 transform, so these are a definition, not a restatement: PARITY UNPINNED.  (dxtlt_oracle_bc7.c) */
void oracle_transform_bc7(const uint8_t *in, uint8_t *out, size_t len);
void oracle_untransform_bc7(const uint8_t *in, uint8_t *out, size_t len);
/* one 16-byte block <-> its 16-byte record (class from byte 0) */
void oracle_bc7_record_of_block(const uint8_t *block, uint8_t *record);
void oracle_bc7_block_of_record(const uint8_t *record, uint8_t *block);
unsigned oracle_bc7_granule(void);
/* force a valid mode marker into byte 0 of every block: mode = (byte 15 & 7) */
void oracle_bc7_force_modes(uint8_t *blocks, size_t len);

/* BC1 block normalisation, the reference's experimental module (dxtlt_oracle_norm.c; PINNED by the reference's unit
 * tests, normalize.rs:505-1076).  mode = ColorNormalizationMode, normalize.rs:487-500. */
enum {
    ORACLE_NORMALIZE_NONE = 0,
    ORACLE_NORMALIZE_COLOR0_ONLY = 1,
    ORACLE_NORMALIZE_REPLICATE_COLOR = 2
};
/* 16 RGBA8888 pixels (64 bytes), row-major: util/bc1_decode.rs:42 */
void oracle_decode_bc1_block(const uint8_t *src, uint8_t *rgba_out);
void oracle_normalize_bc1_blocks(const uint8_t *in, uint8_t *out, size_t len, int mode);            /* in == out allowed */
void oracle_normalize_bc1_split_blocks_in_place(uint8_t *colors, uint8_t *indices, size_t num_blocks, int mode);
int oracle_normalize_bc1_blocks_all_modes(const uint8_t *in, uint8_t *out_none, uint8_t *out_color0,
                                          uint8_t *out_replicate, size_t len);                          /* 1 = any normalised */
int oracle_transform_bc1_with_normalize_blocks(const uint8_t *in, uint8_t *out, size_t len, int mode, int variant,
                                               int split_colour);                                       /* 0 = ok */

/* BC2 / BC3 block normalisation (dxtlt_oracle_norm.c; PINNED by the unit tests of the reference's bc2/bc3
 * experimental/normalize_blocks/normalize.rs).  BC3 AlphaNormalizationMode, bc3 normalize.rs:117-139: */
enum {
    ORACLE_ALPHA_NONE = 0,
    ORACLE_ALPHA_UNIFORM_ALPHA_ZERO_INDICES = 1,
    ORACLE_ALPHA_OPAQUE_FILL_ALL = 2,
    ORACLE_ALPHA_OPAQUE_ZERO_ALPHA_MAX_INDICES = 3
};
void oracle_decode_bc2_block(const uint8_t *src, uint8_t *rgba_out);
void oracle_decode_bc3_block(const uint8_t *src, uint8_t *rgba_out);
/* array forms: kind = 1, 2, 3; 64 bytes of pixels per block */
void oracle_decode_blocks(int kind, const uint8_t *in, uint8_t *rgba_out, size_t num_blocks);
uint64_t oracle_count_pixel_differences(int kind, const uint8_t *a, const uint8_t *b, size_t num_blocks);
void oracle_normalize_bc2_blocks(const uint8_t *in, uint8_t *out, size_t len, int color_mode);
void oracle_normalize_bc2_split_blocks_in_place(const uint8_t *alpha, uint8_t *colors, uint8_t *indices, size_t num_blocks,
                                                int color_mode);
void oracle_normalize_bc2_blocks_all_modes(const uint8_t *in, uint8_t *out_none, uint8_t *out_color0, uint8_t *out_replicate,
                                           size_t len);
void oracle_normalize_bc3_blocks(const uint8_t *in, uint8_t *out, size_t len, int alpha_mode, int color_mode);
void oracle_normalize_bc3_split_blocks_in_place(uint8_t *alpha_endpoints, uint8_t *alpha_indices, uint8_t *color_endpoints,
                                                uint8_t *color_indices, size_t num_blocks, int alpha_mode, int color_mode);
void oracle_normalize_bc3_blocks_all_modes(const uint8_t *in, uint8_t *const outs[12], size_t len);   /* [alpha * 3 + colour] */

#ifdef __cplusplus
}
#endif
#endif
