/*
 * dxtlt_file_formats.h -- the callers on the file side of the hot path: the 4-byte
 * TransformHeader that records how a texture was transformed, and the DDS container handler that uses it.
 * Host-side byte bookkeeping only; the payload itself goes through the gfx950 path (dxtlt_gfx950.h).
 *
 * Reference (paths under /root/reference/src/):
 *   TransformHeader      api/dxt-lossless-transform-file-formats-api/src/embed/mod.rs:105-162
 *                        u32 little-endian: bits 0-3 format, bits 4-31 format data
 *   TransformFormat      .../embed/transform_format.rs:10-31     Bc1=0 Bc2=1 Bc3=2 Bc7=3 Bc6H=4 ...
 *   BC1 / BC2 data       .../embed/formats/bc1.rs:34-82, bc2.rs   version:2 | split_colour:1 | variant:2
 *                        (variant Variant1=0 Variant2=1 Variant3=2 None=3), other bits zero
 *   BC3 data             .../embed/formats/bc3.rs:28-43           upstream: version bits only (placeholder, and
 *                        dispatch refuses BC3).  ADDITIVE here: split_alpha at bit 2, split_colour at bit 3,
 *                        variant at bits 4-5, so that BC3 files written by this build can be restored by it.
 *   DDS parsing          extensions/file-formats/dxt-lossless-transform-dds/src/dds/parse_dds.rs:58-330
 *   C exports            .../dds/exports.rs:12 (is_dds), :39 (parse_dds)
 *   DDS handler          .../handler/file_format_handler.rs:16-135  (copy header, transform payload, copy the
 *                        leftover bytes, overwrite the 'DDS ' magic with the TransformHeader; inverse restores it)
 */
#ifndef DXTLT_FILE_FORMATS_H
#define DXTLT_FILE_FORMATS_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "dlt_size_estimator.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DXTLT_TRANSFORM_HEADER_SIZE 4

/* TransformFormat codes (transform_format.rs:10-31) */
#define DXTLT_TF_BC1 0
#define DXTLT_TF_BC2 1
#define DXTLT_TF_BC3 2
#define DXTLT_TF_BC7 3
/* codes upstream has reserved header layouts for, without a transform behind them yet */
#define DXTLT_TF_BC6H 4
#define DXTLT_TF_RGBA8888 5
#define DXTLT_TF_BGRA8888 6
#define DXTLT_TF_BGR888 7
#define DXTLT_TF_BC4 8
#define DXTLT_TF_BC5 9

/* embed/mod.rs:96-103: bytes a file format must reserve beyond the 4-byte header for these two */
#define DXTLT_BC7_ADDITIONAL_SPACE 48
#define DXTLT_BC6H_ADDITIONAL_SPACE 80

/* status codes of this header's functions */
#define DXTLT_FF_OK 0
#define DXTLT_FF_OUTPUT_TOO_SMALL 1          /* FormatHandlerError::OutputBufferTooSmall */
#define DXTLT_FF_INVALID_INPUT_HEADER 2      /* FormatHandlerError::InvalidInputFileHeader / InvalidRestoredFileHeader */
#define DXTLT_FF_INPUT_TOO_SHORT 3           /* InputTooShort / InputTooShortForStatedTextureSize */
#define DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT 4  /* TransformError::UnknownTransformFormat (not BC1/BC2/BC3) */
#define DXTLT_FF_CORRUPTED_EMBEDDED_DATA 5   /* EmbedError::CorruptedEmbeddedData */
#define DXTLT_FF_INVALID_DATA_ALIGNMENT 6    /* TransformError::InvalidDataAlignment */
#define DXTLT_FF_TRANSFORM_FAILED 7          /* the device path failed; see dxtlt_last_error() */
#define DXTLT_FF_ESTIMATOR_FAILED 8          /* auto mode: estimator callback error */
#define DXTLT_FF_NULL_POINTER 9

/* Pack / unpack a TransformHeader value.  `decorrelation_mode` uses the CORE numbering (None=0, Variant1..3). */
uint32_t dxtlt_transform_header_pack(int32_t transform_format, uint8_t decorrelation_mode,
                                     bool split_alpha_endpoints, bool split_colour_endpoints);
int32_t dxtlt_transform_header_unpack(uint32_t header, int32_t *transform_format, uint8_t *decorrelation_mode,
                                      bool *split_alpha_endpoints, bool *split_colour_endpoints);

/* TransformHeader itself (embed/mod.rs:105-140): format code in bits 0-3, format-specific data in bits 4-31, stored as a
 * little-endian u32.  _new masks `format_data` to 28 bits; _format returns the code, or -1 for one this version does not know
 * (TransformFormat::from_u8 -> None: 10 ... 15), so that files written by later versions are refused, not misread. */
uint32_t dxtlt_transform_header_new(int32_t transform_format, uint32_t format_data);
int32_t dxtlt_transform_header_format(uint32_t header);
uint32_t dxtlt_transform_header_format_data(uint32_t header);
void dxtlt_transform_header_write(uint32_t header, uint8_t *ptr);   /* 4 bytes, little endian */
uint32_t dxtlt_transform_header_read(const uint8_t *ptr);

/* Header data of the formats upstream keeps placeholders for (embed/formats/bc4.rs, bc5.rs: `split_endpoints`; rgba8888.rs,
 * bgra8888.rs, bgr888.rs: `decorrelation`): version:2 (only 0 valid) | flag:1 | reserved:25 (must be zero).  No transform
 * exists for them upstream or here -- the DDS calls answer DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT -- but a tool that walks
 * headers can name them.  _unpack: DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT for any other format code,
 * DXTLT_FF_CORRUPTED_EMBEDDED_DATA for another version or non-zero reserved bits. */
uint32_t dxtlt_transform_header_pack_reserved_format(int32_t transform_format, bool flag);
int32_t dxtlt_transform_header_unpack_reserved_format(uint32_t header, int32_t *transform_format, bool *flag);

/* DdsFormat (parse_dds.rs:6-20), #[repr(u8)] */
enum DdsFormat
#ifdef __cplusplus
  : uint8_t
#endif
{
  NotADds = 0,
  Unknown = 1,
  BC1 = 2,
  BC2 = 3,
  BC3 = 4,
  BC6H = 5,
  BC7 = 6,
  RGBA8888 = 7,
  BGRA8888 = 8,
  BGR888 = 9,
  BC4 = 10,
  BC5 = 11,
};
#ifndef __cplusplus
typedef uint8_t DdsFormat;
#endif

/* parse_dds.rs:22-28, #[repr(C)] */
typedef struct DdsInfo {
  DdsFormat Format;
  uint8_t DataOffset;
  uint32_t DataLength;
} DdsInfo;

/* exports.rs:12 -- at least 128 bytes and the 'DDS ' magic */
bool is_dds(const uint8_t *ptr, size_t len);
/* exports.rs:39 -- {NotADds,0,0} when ptr is NULL, len is 0 or the data is not a DDS */
DdsInfo parse_dds(const uint8_t *ptr, size_t len);

/* DdsHandler::transform_bundle with a manual builder: copy the header, transform the BC1/BC2/BC3 payload with the
 * given settings on the GPU, copy any trailing bytes, replace the magic with the TransformHeader. */
int32_t dxtlt_dds_transform(const uint8_t *input, size_t input_len, uint8_t *output, size_t output_len,
                            uint8_t decorrelation_mode, bool split_alpha_endpoints, bool split_colour_endpoints);
/* Same with an auto builder: settings chosen by transform_bcN_auto with `estimator`. */
int32_t dxtlt_dds_transform_auto(const uint8_t *input, size_t input_len, uint8_t *output, size_t output_len,
                                 const DltSizeEstimator *estimator, bool use_all_decorrelation_modes);
/* DdsHandler::untransform: read the TransformHeader from the first 4 bytes, restore the magic, untransform. */
int32_t dxtlt_dds_untransform(const uint8_t *input, size_t input_len, uint8_t *output, size_t output_len);

/* ADDITIVE, off by default.  Upstream knows no BC7 transform and its dispatch refuses BC7 payloads
 * (handlers/dispatch.rs), which the three functions above reproduce (DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT).  With this
 * switch on (process-wide), BC7 DDS payloads go through this build's own format (dxtlt_bc7.h, docs/BC7_FORMAT.md) and
 * the header carries TransformFormat::Bc7 = 3 (embed/transform_format.rs:18) with the 28 data bits set to a vendor
 * tag (0xD175, bits 27..12) and this build's format version (2, bits 11..0) -- never all zeros, which is what
 * upstream's own first BC7 header version would be; dxtlt_dds_untransform then accepts exactly that word.  Only this
 * build can read such files back. */
void dxtlt_file_formats_enable_bc7(bool enabled);

/* ADDITIVE: many DDS files per call -- the file-after-file loop of the reference's CLI
 * (tools/dxt-lossless-transform-cli/src/commands/transform/mod.rs:154-199) over ONE pinned upload / launch / download
 * pipeline (dxtlt_transform_batch_host, dxtlt_gfx950.h) instead of a PCIe round trip per file.  Every item is checked
 * exactly like dxtlt_dds_transform / dxtlt_dds_untransform would check it and receives that call's status in `status`;
 * the items that pass are transformed together.  inverse = false: `decorrelation_mode` / `split_*` are the settings to
 * apply (as in dxtlt_dds_transform); inverse = true: the settings come from each file's TransformHeader and those fields
 * are ignored.  BC7 files (with the switch above on) ride in the same pipeline.  Returns the number of
 * items whose status is not DXTLT_FF_OK (0 = all done).  If the shared device pipeline fails, every item that was in it
 * gets DXTLT_FF_TRANSFORM_FAILED and its output is unspecified. */
typedef struct DxtltDdsBatchItem {
    const uint8_t *input;
    size_t input_len;
    uint8_t *output;
    size_t output_len;               /* >= input_len */
    uint8_t decorrelation_mode;      /* core numbering; forward only */
    bool split_alpha_endpoints;      /* BC3, forward only */
    bool split_colour_endpoints;     /* forward only */
    int32_t status;                  /* out: DXTLT_FF_* */
} DxtltDdsBatchItem;
size_t dxtlt_dds_transform_batch(DxtltDdsBatchItem *items, size_t count, bool inverse);

#ifdef __cplusplus
}
#endif
#endif /* DXTLT_FILE_FORMATS_H */
