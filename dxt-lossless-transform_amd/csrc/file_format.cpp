// file_format.cpp -- TransformHeader embed/unembed and the DDS container handler (include/dxtlt_file_formats.h).
// Host-side byte bookkeeping; the texture payload goes through the gfx950 path (host_common.h).
// Reference citations for every rule are in the header.
#include "../../include/dxtlt_file_formats.h"

#include <atomic>
#include <cstring>
#include <vector>

#include "../../include/dxtlt_gfx950.h"

#include "host_common.h"
#include "../../include/dxtlt_bc7.h"

namespace {

inline uint32_t rd32(const uint8_t* p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

inline void wr32(uint8_t* p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

// embed/formats/bc1.rs: Variant1=0, Variant2=1, Variant3=2, None=3 in the header; core numbering is None=0, 1, 2, 3
inline uint32_t core_to_header_variant(uint8_t core) { return core == 0 ? 3u : (uint32_t)(core - 1); }
inline uint8_t header_to_core_variant(uint32_t v) { return v == 3 ? 0 : (uint8_t)(v + 1); }

// dds/constants.rs
constexpr uint32_t kDdsMagic = 0x20534444;
constexpr size_t kDdsHeaderSize = 0x80, kDx10HeaderSize = 20;
constexpr size_t kFlagsOff = 0x08, kHeightOff = 0x0C, kWidthOff = 0x10, kMipOff = 0x1C, kPfFlagsOff = 0x50,
                 kFourccOff = 0x54, kBitCountOff = 0x58, kRMaskOff = 0x5C, kGMaskOff = 0x60, kBMaskOff = 0x64,
                 kAMaskOff = 0x68, kDx10FormatOff = 0x80;
constexpr uint32_t kDdsdMipmapCount = 0x20000;
constexpr uint32_t kDdpfAlphaPixels = 0x1, kDdpfAlpha = 0x2, kDdpfFourcc = 0x4, kDdpfRgb = 0x40, kDdpfYuv = 0x200,
                   kDdpfLuminance = 0x20000;

constexpr uint32_t fourcc(char a, char b, char c, char d)
{
    return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24);
}

bool likely_dds(const uint8_t* p, size_t len) { return len >= kDdsHeaderSize && rd32(p) == kDdsMagic; }

// saturating u32 add (the reference uses u32::saturating_add on the running total; the products wrap)
inline uint32_t sat_add(uint32_t a, uint32_t b) { return a + b < a ? 0xFFFFFFFFu : a + b; }

// `levels` more levels of `level_bytes` each, added one by one with saturation -- in closed form.  The mip count comes
// from the file (any u32); once a chain has reached 1 x 1 every further level adds the same few bytes, and a header
// that claims four billion levels must not cost four billion iterations.
inline uint32_t sat_add_repeated(uint32_t total, uint32_t level_bytes, uint32_t levels)
{
    const uint64_t sum = (uint64_t)total + (uint64_t)level_bytes * (uint64_t)levels;
    return sum > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum;
}

bool pixel_length(uint32_t w, uint32_t h, uint32_t mips, uint32_t bpp, uint32_t* out)
{
    if (bpp == 0)
        return false;
    uint32_t total = 0;
    for (uint32_t i = 0; i < mips; ++i) {
        if (w == 1 && h == 1) {
            total = sat_add_repeated(total, bpp, mips - i);
            break;
        }
        total = sat_add(total, w * h * bpp);
        w = w / 2 > 1 ? w / 2 : 1;
        h = h / 2 > 1 ? h / 2 : 1;
    }
    *out = total;
    return true;
}

bool block_length(uint8_t fmt, uint32_t w, uint32_t h, uint32_t mips, uint32_t* out)
{
    const uint32_t block = (fmt == BC1 || fmt == BC4) ? 8 : 16;
    uint32_t total = 0;
    for (uint32_t i = 0; i < mips; ++i) {
        if (w == 1 && h == 1) {
            total = sat_add_repeated(total, block, mips - i);
            break;
        }
        const uint32_t bw = w / 4 + (w % 4 != 0), bh = h / 4 + (h % 4 != 0);
        total = sat_add(total, bw * bh * block);
        w = w / 2 > 1 ? w / 2 : 1;
        h = h / 2 > 1 ? h / 2 : 1;
    }
    *out = total;
    return true;
}

uint8_t detect_uncompressed(const uint8_t* d)
{
    const uint32_t flags = rd32(d + kPfFlagsOff), bits = rd32(d + kBitCountOff);
    const uint32_t r = rd32(d + kRMaskOff), g = rd32(d + kGMaskOff), b = rd32(d + kBMaskOff), a = rd32(d + kAMaskOff);
    if (bits == 24)
        return (r == 0x00FF0000 && g == 0x0000FF00 && b == 0x000000FF && a == 0) ? BGR888 : Unknown;
    if (bits == 32 && (flags & kDdpfAlphaPixels)) {
        if (r == 0x000000FF && g == 0x0000FF00 && b == 0x00FF0000 && a == 0xFF000000)
            return RGBA8888;
        if (r == 0x00FF0000 && g == 0x0000FF00 && b == 0x000000FF && a == 0xFF000000)
            return BGRA8888;
    }
    return Unknown;
}

// parse_dds_ignore_magic (parse_dds.rs:79-); returns false for "None"
bool parse_ignore_magic(const uint8_t* d, size_t len, DdsInfo* info)
{
    if (len < kDdsHeaderSize)
        return false;
    const uint32_t cc = rd32(d + kFourccOff);
    uint8_t fmt;
    size_t off;
    if (cc == fourcc('D', 'X', '1', '0')) {
        if (len < kDdsHeaderSize + kDx10HeaderSize)
            return false;
        const uint32_t dxgi = rd32(d + kDx10FormatOff);
        if (dxgi >= 70 && dxgi <= 72) fmt = BC1;
        else if (dxgi >= 73 && dxgi <= 75) fmt = BC2;
        else if (dxgi >= 76 && dxgi <= 78) fmt = BC3;
        else if (dxgi >= 79 && dxgi <= 81) fmt = BC4;
        else if (dxgi >= 82 && dxgi <= 84) fmt = BC5;
        else if (dxgi >= 94 && dxgi <= 96) fmt = BC6H;
        else if (dxgi >= 97 && dxgi <= 99) fmt = BC7;
        else if (dxgi >= 27 && dxgi <= 32) fmt = RGBA8888;
        else if (dxgi == 87 || dxgi == 90 || dxgi == 91) fmt = BGRA8888;
        else fmt = Unknown;
        off = kDdsHeaderSize + kDx10HeaderSize;
    } else {
        const uint32_t pf = rd32(d + kPfFlagsOff);
        if (pf & kDdpfFourcc) {
            if (cc == fourcc('D', 'X', 'T', '1')) fmt = BC1;
            else if (cc == fourcc('D', 'X', 'T', '2') || cc == fourcc('D', 'X', 'T', '3')) fmt = BC2;
            else if (cc == fourcc('D', 'X', 'T', '4') || cc == fourcc('D', 'X', 'T', '5')) fmt = BC3;
            else if (cc == fourcc('B', 'C', '4', 'U') || cc == fourcc('B', 'C', '4', 'S') || cc == fourcc('A', 'T', 'I', '1')) fmt = BC4;
            else if (cc == fourcc('B', 'C', '5', 'U') || cc == fourcc('B', 'C', '5', 'S') || cc == fourcc('A', 'T', 'I', '2')) fmt = BC5;
            else fmt = Unknown;
        } else if (pf & kDdpfRgb) {
            fmt = detect_uncompressed(d);
        } else {
            fmt = Unknown;
        }
        off = kDdsHeaderSize;
    }

    // calculate_data_length (parse_dds.rs:190-)
    const uint32_t flags = rd32(d + kFlagsOff), height = rd32(d + kHeightOff), width = rd32(d + kWidthOff);
    const uint32_t raw_mips = rd32(d + kMipOff);
    const uint32_t mips = (flags & kDdsdMipmapCount) ? (raw_mips > 1 ? raw_mips : 1) : 1;
    uint32_t length = 0;
    bool ok = true;
    switch (fmt) {
    case BC1: case BC2: case BC3: case BC4: case BC5: case BC6H: case BC7:
        ok = block_length(fmt, width, height, mips, &length);
        break;
    case RGBA8888: case BGRA8888:
        ok = pixel_length(width, height, mips, 4, &length);
        break;
    case BGR888:
        ok = pixel_length(width, height, mips, 3, &length);
        break;
    default: {  // Unknown: try the pixel format (calculate_uncompressed_data_length)
        const uint32_t pf = rd32(d + kPfFlagsOff), bits = rd32(d + kBitCountOff);
        if ((pf & (kDdpfRgb | kDdpfLuminance | kDdpfYuv | kDdpfAlpha)) == 0 || bits % 8 != 0)
            ok = false;
        else
            ok = pixel_length(width, height, mips, bits / 8, &length);
        break;
    }
    }
    info->Format = (DdsFormat)fmt;
    info->DataOffset = (uint8_t)off;
    info->DataLength = ok ? length : 0;  // unwrap_or(0)
    return true;
}

int32_t map_device_status(int32_t st)
{
    if (st == dxtlt_host::kOk) return DXTLT_FF_OK;
    if (st == dxtlt_host::kInvalidLength) return DXTLT_FF_INVALID_DATA_ALIGNMENT;
    if (st == dxtlt_host::kEstimator || st == dxtlt_host::kAllocation) return DXTLT_FF_ESTIMATOR_FAILED;
    return DXTLT_FF_TRANSFORM_FAILED;
}

// format_conversion.rs: only BC1/BC2 upstream; BC3 additive here
int dds_to_bcn(uint8_t fmt) { return fmt == BC1 ? 1 : fmt == BC2 ? 2 : fmt == BC3 ? 3 : 0; }

// BC7 payloads: refused as upstream (dispatch.rs knows no BC7 transform) unless the caller opts in to this build's own
// format (include/dxtlt_bc7.h).  TransformFormat::Bc7 = 3 exists upstream and its 28 data bits are unassigned ("3 bits
// for each of BC7's 8 modes" is a comment, embed/mod.rs:49-51), so all-zero data bits are what upstream's first BC7
// header version would look like.  Files written here must not be mistaken for that: the data bits carry a vendor tag
// in the upper 16 bits and this build's format version in the lower 12, and anything else -- all zeros included -- is
// refused on the way back.
std::atomic<bool> g_bc7_enabled{false};
constexpr uint32_t kBc7VendorTag = 0xD175u;    // data bits 27..12
constexpr uint32_t kBc7FormatVersion = 2u;     // data bits 11..0: docs/BC7_FORMAT.md version
constexpr uint32_t kBc7PrivateHeader = (uint32_t)DXTLT_TF_BC7 | (((kBc7VendorTag << 12) | kBc7FormatVersion) << 4);

int32_t dds_transform_common(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len,
                             const DltSizeEstimator* estimator, bool use_all, uint8_t mode, bool sa, bool sc)
{
    if (input == nullptr || output == nullptr)
        return DXTLT_FF_NULL_POINTER;
    if (output_len < input_len)
        return DXTLT_FF_OUTPUT_TOO_SMALL;
    DdsInfo info;
    if (!likely_dds(input, input_len) || !parse_ignore_magic(input, input_len, &info))
        return DXTLT_FF_INVALID_INPUT_HEADER;
    const size_t off = info.DataOffset, length = info.DataLength;
    if (input_len < off + length)
        return DXTLT_FF_INPUT_TOO_SHORT;
    if (info.Format == BC7 && g_bc7_enabled.load(std::memory_order_relaxed)) {
        // no settings and nothing to estimate: the format has one layout
        if (length % 16 != 0)
            return DXTLT_FF_INVALID_DATA_ALIGNMENT;
        std::memcpy(output, input, off);
        const int32_t st7 = dxtlt_transform_bc7(input + off, output + off, length);
        if (st7 != dxtlt_host::kOk)
            return map_device_status(st7);
        if (input_len > off + length)
            std::memcpy(output + off + length, input + off + length, input_len - off - length);
        wr32(output, kBc7PrivateHeader);
        return DXTLT_FF_OK;
    }
    const int bcn = dds_to_bcn(info.Format);
    if (bcn == 0)
        return DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT;
    if (length % (bcn == 1 ? 8 : 16) != 0)
        return DXTLT_FF_INVALID_DATA_ALIGNMENT;
    if (bcn != 3)
        sa = false;

    std::memcpy(output, input, off);
    int32_t st;
    if (estimator != nullptr) {
        dxtlt_host::AutoChoice c{};
        st = dxtlt_host::transform_auto(bcn, input + off, output + off, length, estimator, use_all, &c);
        mode = c.mode;
        sa = c.split_alpha;
        sc = c.split_colour;
    } else {
        if (mode > 3)
            return DXTLT_FF_CORRUPTED_EMBEDDED_DATA;
        st = dxtlt_host::transform(bcn, false, input + off, output + off, length, mode, sa, sc);
    }
    if (st != dxtlt_host::kOk)
        return map_device_status(st);
    if (input_len > off + length)
        std::memcpy(output + off + length, input + off + length, input_len - off - length);
    wr32(output, dxtlt_transform_header_pack(bcn - 1, mode, sa, sc));  // overwrites the 'DDS ' magic
    return DXTLT_FF_OK;
}

}  // namespace

extern "C" {

uint32_t dxtlt_transform_header_pack(int32_t transform_format, uint8_t mode, bool sa, bool sc)
{
    uint32_t data = 0;  // header_version 0 in bits 0-1
    if (transform_format == DXTLT_TF_BC1 || transform_format == DXTLT_TF_BC2) {
        data |= (sc ? 1u : 0u) << 2;
        data |= core_to_header_variant(mode & 3) << 3;
    } else if (transform_format == DXTLT_TF_BC3) {  // additive layout, see the header file
        data |= (sa ? 1u : 0u) << 2;
        data |= (sc ? 1u : 0u) << 3;
        data |= core_to_header_variant(mode & 3) << 4;
    }
    return ((uint32_t)transform_format & 0xF) | (data << 4);
}

int32_t dxtlt_transform_header_unpack(uint32_t header, int32_t* transform_format, uint8_t* mode, bool* sa, bool* sc)
{
    const uint32_t fmt = header & 0xF, data = header >> 4;
    if ((data & 3) != 0)
        return DXTLT_FF_CORRUPTED_EMBEDDED_DATA;  // unknown header version
    uint8_t m = 0;
    bool a = false, c = false;
    if (fmt == DXTLT_TF_BC1 || fmt == DXTLT_TF_BC2) {
        c = (data >> 2) & 1;
        m = header_to_core_variant((data >> 3) & 3);
    } else if (fmt == DXTLT_TF_BC3) {
        a = (data >> 2) & 1;
        c = (data >> 3) & 1;
        m = header_to_core_variant((data >> 4) & 3);
    } else {
        return DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT;
    }
    if (transform_format) *transform_format = (int32_t)fmt;
    if (mode) *mode = m;
    if (sa) *sa = a;
    if (sc) *sc = c;
    return DXTLT_FF_OK;
}

uint32_t dxtlt_transform_header_new(int32_t transform_format, uint32_t format_data)
{
    return ((uint32_t)transform_format & 0xFu) | (format_data << 4);
}

int32_t dxtlt_transform_header_format(uint32_t header)
{
    const uint32_t code = header & 0xFu;
    return code <= DXTLT_TF_BC5 ? (int32_t)code : -1;
}

uint32_t dxtlt_transform_header_format_data(uint32_t header) { return header >> 4; }

void dxtlt_transform_header_write(uint32_t header, uint8_t* ptr)
{
    if (ptr != nullptr)
        wr32(ptr, header);
}

uint32_t dxtlt_transform_header_read(const uint8_t* ptr) { return ptr != nullptr ? rd32(ptr) : 0; }

namespace {
bool is_reserved_flag_format(uint32_t code)
{
    return code == DXTLT_TF_RGBA8888 || code == DXTLT_TF_BGRA8888 || code == DXTLT_TF_BGR888 || code == DXTLT_TF_BC4 ||
           code == DXTLT_TF_BC5;
}
}  // namespace

uint32_t dxtlt_transform_header_pack_reserved_format(int32_t transform_format, bool flag)
{
    return dxtlt_transform_header_new(transform_format, (flag ? 1u : 0u) << 2);   // version 0, reserved 0
}

int32_t dxtlt_transform_header_unpack_reserved_format(uint32_t header, int32_t* transform_format, bool* flag)
{
    const uint32_t code = header & 0xFu, data = header >> 4;
    if (!is_reserved_flag_format(code))
        return DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT;
    if ((data & 3u) != 0 || (data >> 3) != 0)
        return DXTLT_FF_CORRUPTED_EMBEDDED_DATA;
    if (transform_format) *transform_format = (int32_t)code;
    if (flag) *flag = ((data >> 2) & 1u) != 0;
    return DXTLT_FF_OK;
}

void dxtlt_file_formats_enable_bc7(bool enabled) { g_bc7_enabled.store(enabled, std::memory_order_relaxed); }

bool is_dds(const uint8_t* ptr, size_t len)
{
    if (ptr == nullptr || len == 0)
        return false;
    return likely_dds(ptr, len);
}

DdsInfo parse_dds(const uint8_t* ptr, size_t len)
{
    DdsInfo none;
    none.Format = (DdsFormat)NotADds;
    none.DataOffset = 0;
    none.DataLength = 0;
    if (ptr == nullptr || len == 0 || !likely_dds(ptr, len))
        return none;
    DdsInfo info = none;
    if (!parse_ignore_magic(ptr, len, &info))
        return none;
    return info;
}

int32_t dxtlt_dds_transform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len, uint8_t mode,
                            bool sa, bool sc)
{
    return dds_transform_common(input, input_len, output, output_len, nullptr, false, mode, sa, sc);
}

int32_t dxtlt_dds_transform_auto(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len,
                                 const DltSizeEstimator* estimator, bool use_all)
{
    if (estimator == nullptr)
        return DXTLT_FF_NULL_POINTER;
    return dds_transform_common(input, input_len, output, output_len, estimator, use_all, 0, false, false);
}

int32_t dxtlt_dds_untransform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len)
{
    if (input == nullptr || output == nullptr)
        return DXTLT_FF_NULL_POINTER;
    if (input_len < DXTLT_TRANSFORM_HEADER_SIZE)
        return DXTLT_FF_INPUT_TOO_SHORT;
    if (output_len < input_len)
        return DXTLT_FF_OUTPUT_TOO_SMALL;
    const uint32_t header = rd32(input);
    DdsInfo info;
    if (!parse_ignore_magic(input, input_len, &info))
        return DXTLT_FF_INVALID_INPUT_HEADER;
    const size_t off = info.DataOffset, length = info.DataLength;
    if (input_len < off + length)
        return DXTLT_FF_INPUT_TOO_SHORT;

    int32_t tf = 0;
    uint8_t mode = 0;
    bool sa = false, sc = false;
    // dispatch_untransform (handlers/dispatch.rs:39-): format first, then the details, then the alignment
    if ((header & 0xF) == DXTLT_TF_BC7 && g_bc7_enabled.load(std::memory_order_relaxed)) {
        if (header != kBc7PrivateHeader)
            return DXTLT_FF_CORRUPTED_EMBEDDED_DATA;  // not this build's tag + version (all-zero data bits: upstream's to assign)
        if (length % 16 != 0)
            return DXTLT_FF_INVALID_DATA_ALIGNMENT;
        wr32(output, kDdsMagic);
        std::memcpy(output + 4, input + 4, off - 4);
        const int32_t st7 = dxtlt_untransform_bc7(input + off, output + off, length);
        if (st7 != dxtlt_host::kOk)
            return map_device_status(st7);
        if (input_len > off + length)
            std::memcpy(output + off + length, input + off + length, input_len - off - length);
        return DXTLT_FF_OK;
    }
    if ((header & 0xF) > DXTLT_TF_BC3)
        return DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT;
    int32_t rc = dxtlt_transform_header_unpack(header, &tf, &mode, &sa, &sc);
    if (rc != DXTLT_FF_OK)
        return rc;
    const int bcn = tf + 1;
    if (length % (bcn == 1 ? 8 : 16) != 0)
        return DXTLT_FF_INVALID_DATA_ALIGNMENT;

    wr32(output, kDdsMagic);
    std::memcpy(output + 4, input + 4, off - 4);
    int32_t st = dxtlt_host::transform(bcn, true, input + off, output + off, length, mode, sa, sc);
    if (st != dxtlt_host::kOk)
        return map_device_status(st);
    if (input_len > off + length)
        std::memcpy(output + off + length, input + off + length, input_len - off - length);
    return DXTLT_FF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Many DDS files per call: the file-after-file loop of the reference's CLI (tools/dxt-lossless-transform-cli/src/
// commands/transform/mod.rs:154-199) with ONE upload / launch / download pipeline under it (dxtlt_transform_batch_host)
// instead of a PCIe round trip per file.  Every item is checked exactly like the single-file calls and gets their
// status; items that pass have their header and trailing bytes copied, their payloads go through the batch together,
// and their TransformHeader (or 'DDS ' magic) is written once the batch has succeeded.  BC7 files (with the switch on)
// ride in the same batch as format 7.
// ---------------------------------------------------------------------------------------------------------------
size_t dxtlt_dds_transform_batch(DxtltDdsBatchItem* items, size_t count, bool inverse)
{
    if (items == nullptr)
        return count;
    struct Pending {
        size_t item;
        uint32_t first_word;   // what goes into bytes 0..3 of the output once the payload is done
    };
    std::vector<DxtltBatchItem> batch;
    std::vector<Pending> pending;
    size_t failed = 0;
    for (size_t i = 0; i < count; ++i) {
        DxtltDdsBatchItem& it = items[i];
        auto reject = [&](int32_t st) {
            it.status = st;
            ++failed;
        };
        if (it.input == nullptr || it.output == nullptr) { reject(DXTLT_FF_NULL_POINTER); continue; }
        if (inverse && it.input_len < DXTLT_TRANSFORM_HEADER_SIZE) { reject(DXTLT_FF_INPUT_TOO_SHORT); continue; }
        if (it.output_len < it.input_len) { reject(DXTLT_FF_OUTPUT_TOO_SMALL); continue; }
        DdsInfo info;
        if ((!inverse && !likely_dds(it.input, it.input_len)) || !parse_ignore_magic(it.input, it.input_len, &info)) {
            reject(DXTLT_FF_INVALID_INPUT_HEADER);
            continue;
        }
        const size_t off = info.DataOffset, length = info.DataLength;
        if (it.input_len < off + length) { reject(DXTLT_FF_INPUT_TOO_SHORT); continue; }
        int bcn;
        uint8_t mode = it.decorrelation_mode;
        bool sa = it.split_alpha_endpoints, sc = it.split_colour_endpoints;
        uint32_t first_word;
        if (!inverse) {
            bcn = dds_to_bcn(info.Format);
            if (info.Format == BC7 && g_bc7_enabled.load(std::memory_order_relaxed)) {
                bcn = 7;   // opt-in: this build's own format, no settings; rides in the same batch
                if (length % 16 != 0) { reject(DXTLT_FF_INVALID_DATA_ALIGNMENT); continue; }
                mode = 0;
                sa = sc = false;
                first_word = kBc7PrivateHeader;
            } else {
                if (bcn == 0) { reject(DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT); continue; }
                if (length % (bcn == 1 ? 8 : 16) != 0) { reject(DXTLT_FF_INVALID_DATA_ALIGNMENT); continue; }
                if (mode > 3) { reject(DXTLT_FF_CORRUPTED_EMBEDDED_DATA); continue; }
                if (bcn != 3)
                    sa = false;
                first_word = dxtlt_transform_header_pack(bcn - 1, mode, sa, sc);
            }
        } else {
            const uint32_t header = rd32(it.input);
            if ((header & 0xF) == DXTLT_TF_BC7 && g_bc7_enabled.load(std::memory_order_relaxed)) {
                if (header != kBc7PrivateHeader) { reject(DXTLT_FF_CORRUPTED_EMBEDDED_DATA); continue; }
                if (length % 16 != 0) { reject(DXTLT_FF_INVALID_DATA_ALIGNMENT); continue; }
                bcn = 7;
                mode = 0;
                sa = sc = false;
            } else {
                if ((header & 0xF) > DXTLT_TF_BC3) { reject(DXTLT_FF_UNKNOWN_TRANSFORM_FORMAT); continue; }
                int32_t tf = 0;
                const int32_t rc = dxtlt_transform_header_unpack(header, &tf, &mode, &sa, &sc);
                if (rc != DXTLT_FF_OK) { reject(rc); continue; }
                bcn = tf + 1;
                if (length % (bcn == 1 ? 8 : 16) != 0) { reject(DXTLT_FF_INVALID_DATA_ALIGNMENT); continue; }
            }
            first_word = kDdsMagic;
        }
        // everything around the payload; bytes 0..3 stay the input's until the payload has been transformed
        std::memcpy(it.output, it.input, off);
        if (it.input_len > off + length)
            std::memcpy(it.output + off + length, it.input + off + length, it.input_len - off - length);
        it.status = DXTLT_FF_OK;
        pending.push_back({i, first_word});
        if (length != 0) {
            DxtltBatchItem b{};
            b.d_input = it.input + off;
            b.d_output = it.output + off;
            b.len = length;
            b.format = (uint8_t)bcn;
            b.inverse = inverse ? 1 : 0;
            b.decorrelation_mode = mode;
            b.split_alpha_endpoints = sa ? 1 : 0;
            b.split_colour_endpoints = sc ? 1 : 0;
            batch.push_back(b);
        }
    }
    const int32_t st = batch.empty() ? dxtlt_host::kOk : dxtlt_transform_batch_host(batch.data(), batch.size());
    for (const Pending& p : pending) {
        if (st == dxtlt_host::kOk) {
            wr32(items[p.item].output, p.first_word);
        } else {
            items[p.item].status = map_device_status(st);
            ++failed;
        }
    }
    return failed;
}

}  // extern "C"
