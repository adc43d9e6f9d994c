// C++ host API (include/dxt_lossless_transform.hpp) test program, driven by tests/test_cpp_api.py.
//   test_cpp_api cpu   -- validation paths only (no device needed)
//   test_cpp_api gpu   -- round trips on the device, reference-generator inputs
// The reference's own tests this mirrors: bc1 transform/safe/transform_with_settings.rs tests (:225-330),
// bc1-api transform/manual_transform_builder.rs and auto_transform_builder.rs tests.
#include <cstdio>
#include <cstring>
#include <atomic>
#include <vector>

#include "../../include/dxt_lossless_transform.hpp"

using namespace dxt_lossless_transform;

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);     \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

// reference generator for BC1 (bc1 test_prelude.rs:81-105)
static std::vector<uint8_t> gen_bc1(size_t blocks)
{
    std::vector<uint8_t> d(blocks * 8);
    uint8_t c = 0, i = 128;
    for (size_t b = 0; b < blocks; ++b) {
        for (int k = 0; k < 4; ++k) {
            d[b * 8 + k] = (uint8_t)(c + k);
            d[b * 8 + 4 + k] = (uint8_t)(i + k);
        }
        c = (uint8_t)(c + 4);
        i = (uint8_t)(i + 4);
    }
    return d;
}

struct DummyEstimator {  // bc1 test_prelude.rs:44-62: max 0, estimate = len
    int calls = 0;
    bool max_compressed_size(size_t, size_t& out) { out = 0; return true; }
    bool estimate_compressed_size(const uint8_t*, size_t len, uint8_t*, size_t, size_t& out) { ++calls; out = len; return true; }
};
struct SharedCounterEstimator {  // thread-safe: for dxtlt_set_auto_estimator_threads
    std::atomic<int>* calls;
    bool max_compressed_size(size_t len, size_t& out) { out = len; return true; }
    bool estimate_compressed_size(const uint8_t* p, size_t len, uint8_t*, size_t, size_t& out)
    {
        calls->fetch_add(1);
        size_t h = 0;
        for (size_t i = 0; i < len; ++i) h = h * 131 + p[i];
        out = h % 100000;   // depends on every byte shown
        return true;
    }
};
struct FailingEstimator {  // bc1 transform/mod.rs:120-138
    bool max_compressed_size(size_t, size_t& out) { out = 0; return true; }
    bool estimate_compressed_size(const uint8_t*, size_t, uint8_t*, size_t, size_t&) { return false; }
};

static void cpu_tests()
{
    // defaults and combinations
    core::Bc1TransformSettings d1;
    CHECK(d1.decorrelation_mode == core::YCoCgVariant::Variant1 && d1.split_colour_endpoints);
    core::Bc3TransformSettings d3;
    CHECK(d3.split_alpha_endpoints && d3.split_colour_endpoints);
    CHECK(core::Bc1TransformSettings::all_combinations().size() == 8);
    CHECK(core::Bc3TransformSettings::all_combinations().size() == 16);
    CHECK((int)api::YCoCgVariant::None == 3 && (int)core::YCoCgVariant::None == 0);
    CHECK(api::to_internal_variant(api::YCoCgVariant::Variant2) == core::YCoCgVariant::Variant2);
    CHECK(api::from_internal_variant(core::YCoCgVariant::None) == api::YCoCgVariant::None);

    // safe wrappers: length first, then size (no device touched)
    std::vector<uint8_t> in(24), out(24);
    auto e = core::transform_bc1_with_settings_safe(in.data(), 12, out.data(), 24, {});
    CHECK(e.kind == core::ValidationError::InvalidLength && e.length == 12);
    e = core::transform_bc1_with_settings_safe(in.data(), 24, out.data(), 16, {});
    CHECK(e.kind == core::ValidationError::OutputBufferTooSmall && e.needed == 24 && e.actual == 16);
    e = core::untransform_bc3_with_settings_safe(in.data(), 24, out.data(), 1, {});
    CHECK(e.kind == core::ValidationError::InvalidLength);
    auto be = api::Bc1ManualTransformBuilder().transform(in.data(), 20, out.data(), 24);
    CHECK(be.kind == api::Error::InvalidLength);
    be = api::Bc2ManualTransformBuilder().untransform(in.data(), 16, out.data(), 8);
    CHECK(be.kind == api::Error::OutputBufferTooSmall);
    // builder setters are value-returning, like the Rust builder
    auto b = api::Bc1ManualTransformBuilder().decorrelation_mode(api::YCoCgVariant::None).split_colour_endpoints(false);
    CHECK(b.settings().decorrelation_mode == core::YCoCgVariant::None && !b.settings().split_colour_endpoints);
    // zero-length input is Ok and needs no device
    CHECK(core::transform_bc1_with_settings_safe(in.data(), 0, out.data(), 0, {}).is_ok());
    // auto builder validates before estimating
    api::Bc1AutoTransformBuilder<DummyEstimator> ab{DummyEstimator{}};
    CHECK(ab.transform(in.data(), 20, out.data(), 24).second.kind == api::Error::InvalidLength);
}

static void gpu_tests()
{
    const size_t blocks = 3001;
    std::vector<uint8_t> x = gen_bc1(blocks), y(x.size()), z(x.size());
    for (auto s : core::Bc1TransformSettings::all_combinations()) {
        CHECK(core::transform_bc1_with_settings_safe(x.data(), x.size(), y.data(), y.size(), s).is_ok());
        CHECK(core::untransform_bc1_with_settings_safe(y.data(), y.size(), z.data(), z.size(), s).is_ok());
        CHECK(z == x);
        // index stream is verbatim in the second half for every setting
        bool idx_ok = true;
        for (size_t b = 0; b < blocks && idx_ok; ++b)
            idx_ok = std::memcmp(&y[4 * blocks + 4 * b], &x[8 * b + 4], 4) == 0;
        CHECK(idx_ok);
    }
    // hand-derived vector: generator n=3, Variant1 + split
    std::vector<uint8_t> g = gen_bc1(3), t(24);
    core::transform_bc1_with_settings(g.data(), t.data(), 24, {});
    const uint8_t want[24] = {0x04, 0x10, 0x02, 0x9f, 0x50, 0xe6, 0x9b, 0xf7, 0x89, 0xbe, 0xd7, 0x05,
                              0x80, 0x81, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x8b};
    CHECK(std::memcmp(t.data(), want, 24) == 0);

    // BC2 / BC3 round trips through the pointer API
    std::vector<uint8_t> w(16 * 1777), w2(w.size()), w3(w.size());
    for (size_t i = 0; i < w.size(); ++i) w[i] = (uint8_t)(i * 131 + (i >> 7));
    for (auto s : core::Bc3TransformSettings::all_combinations()) {
        core::transform_bc3_with_settings(w.data(), w2.data(), w.size(), s);
        core::untransform_bc3_with_settings(w2.data(), w3.data(), w.size(), s);
        CHECK(w3 == w);
    }
    for (auto s : core::Bc2TransformSettings::all_combinations()) {
        core::transform_bc2_with_settings(w.data(), w2.data(), w.size(), s);
        core::untransform_bc2_with_settings(w2.data(), w3.data(), w.size(), s);
        CHECK(w3 == w);
    }

    // builders (stable API)
    auto mb = api::Bc1ManualTransformBuilder().decorrelation_mode(api::YCoCgVariant::Variant3).split_colour_endpoints(false);
    CHECK(mb.transform(x.data(), x.size(), y.data(), y.size()).is_ok());
    CHECK(mb.untransform(y.data(), y.size(), z.data(), z.size()).is_ok());
    CHECK(z == x);

    // auto: constant estimator -> strict '<' keeps the first candidate (None, no split); 4 candidates tried
    api::Bc1AutoTransformBuilder<DummyEstimator> ab{DummyEstimator{}};
    auto r = ab.transform(x.data(), x.size(), y.data(), y.size());
    CHECK(r.second.is_ok());
    CHECK(r.first.settings().decorrelation_mode == core::YCoCgVariant::None && !r.first.settings().split_colour_endpoints);
    CHECK(r.first.untransform(y.data(), y.size(), z.data(), z.size()).is_ok());
    CHECK(z == x);
    auto ultra = api::Bc2AutoTransformBuilder<DummyEstimator>::new_ultra(DummyEstimator{});
    CHECK(ultra.transform(w.data(), w.size(), w2.data(), w2.size()).second.is_ok());
    api::Bc1AutoTransformBuilder<FailingEstimator> fb{FailingEstimator{}};
    CHECK(fb.transform(x.data(), x.size(), y.data(), y.size()).second.kind == api::Error::SizeEstimationFailed);
    core::EstimateSettings<DummyEstimator> es{DummyEstimator{}, true};
    auto r3 = core::transform_bc3_auto(w.data(), w2.data(), w.size(), es);
    CHECK(r3.second.is_ok() && es.size_estimator.calls == 32);  // 16 candidates x (alpha + colour endpoints)
    {
    // opt-in parallel estimator: the same choice and bytes, every distinct section estimated once (8 colour + 2 alpha)
    std::atomic<int> calls_seq{0}, calls_par{0};
    std::vector<uint8_t> out_seq(w.size()), out_par(w.size());
    core::EstimateSettings<SharedCounterEstimator> e1{SharedCounterEstimator{&calls_seq}, true}, e4{SharedCounterEstimator{&calls_par}, true};
    auto seq = core::transform_bc3_auto(w.data(), out_seq.data(), w.size(), e1);
    core::set_auto_estimator_threads(4);
    CHECK(core::auto_estimator_threads() == 4);
    auto par = core::transform_bc3_auto(w.data(), out_par.data(), w.size(), e4);
    core::set_auto_estimator_threads(1);
    CHECK(seq.second.is_ok() && par.second.is_ok() && calls_seq.load() == 32 && calls_par.load() == 10 && out_seq == out_par);
    CHECK(seq.first.decorrelation_mode == par.first.decorrelation_mode && seq.first.split_alpha_endpoints == par.first.split_alpha_endpoints &&
          seq.first.split_colour_endpoints == par.first.split_colour_endpoints);
    }

    // experimental block normalisation: the reference's unit vectors (normalize.rs:508-598, 750-848)
    namespace ex = core::experimental;
    const uint8_t solid[8] = {0x00, 0xF8, 0x01, 0x01, 0, 0, 0, 0}, clear[8] = {0x00, 0x80, 0x00, 0xF8, 0xFF, 0xFF, 0xFF, 0xFF};
    uint8_t two[16], n0[16], n1[16], n2[16];
    std::memcpy(two, solid, 8);
    std::memcpy(two + 8, clear, 8);
    ex::normalize_blocks(two, n1, 16, ex::ColorNormalizationMode::Color0Only);
    const uint8_t want_c0[16] = {0x00, 0xF8, 0, 0, 0, 0, 0, 0, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
    CHECK(std::memcmp(n1, want_c0, 16) == 0);
    CHECK(ex::normalize_blocks_all_modes(two, {n0, n1, n2}, 16));
    const uint8_t want_rep[8] = {0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0};
    CHECK(std::memcmp(n0, solid, 8) == 0 && std::memcmp(n1, want_c0, 8) == 0 && std::memcmp(n2, want_rep, 8) == 0);
    CHECK(std::memcmp(n0 + 8, want_c0 + 8, 8) == 0 && std::memcmp(n2 + 8, want_c0 + 8, 8) == 0);
    ex::normalize_blocks(two, two, 16, ex::ColorNormalizationMode::ReplicateColor);   // in place
    CHECK(std::memcmp(two, want_rep, 8) == 0);
    // fused normalise + transform == transform of the normalised blocks; the details convert to untransform settings
    ex::Bc1TransformDetailsWithNormalization det{ex::ColorNormalizationMode::Color0Only, core::YCoCgVariant::Variant2, true};
    std::vector<uint8_t> xs = gen_bc1(2049), xn(xs.size()), f1(xs.size()), f2(xs.size()), back(xs.size());
    for (size_t b = 0; b < 2049; b += 3) std::memset(&xs[8 * b + 4], 0, 4);   // every third block solid
    ex::normalize_blocks(xs.data(), xn.data(), xs.size(), det.color_normalization_mode);
    CHECK(xn != xs);
    ex::transform_bc1_with_normalize_blocks(xs.data(), f1.data(), nullptr, xs.size(), det);
    core::transform_bc1_with_settings(xn.data(), f2.data(), xn.size(), det);
    CHECK(f1 == f2);
    core::untransform_bc1_with_settings(f1.data(), back.data(), f1.size(), det);
    CHECK(back == xn);
    core::EstimateSettings<DummyEstimator> en{DummyEstimator{}, false};
    auto rn = ex::transform_bc1_auto_with_normalization(xs.data(), f1.data(), xs.size(), en);
    CHECK(rn.second.is_ok() && en.size_estimator.calls == 12);   // 3 modes x 4 candidates
    CHECK(rn.first == ex::Bc1TransformDetailsWithNormalization(ex::ColorNormalizationMode::None, core::YCoCgVariant::None, false));

    // BC2 / BC3 normalisation: the reference's unit vectors (bc2 / bc3 normalize.rs tests)
    const uint8_t b2[16] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x00, 0xF8, 0x01, 0x01, 0, 0, 0, 0};
    uint8_t o2[16];
    ex::bc2::normalize_blocks(b2, o2, 16, ex::bc2::ColorNormalizationMode::ReplicateColor);
    const uint8_t want2[16] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0};
    CHECK(std::memcmp(o2, want2, 16) == 0);
    const uint8_t b3[16] = {0xFF, 0xFF, 0, 0, 0, 0, 0, 0, 0x00, 0xF8, 0x12, 0x34, 0, 0, 0, 0};
    uint8_t o3[16];
    ex::bc3::normalize_blocks(b3, o3, 16, ex::bc3::AlphaNormalizationMode::OpaqueZeroAlphaMaxIndices,
                              ex::bc3::ColorNormalizationMode::Color0Only);
    const uint8_t want3[16] = {0, 0, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x00, 0xF8, 0, 0, 0, 0, 0, 0};
    CHECK(std::memcmp(o3, want3, 16) == 0);
    std::vector<std::vector<uint8_t>> outs(12, std::vector<uint8_t>(16));
    std::array<std::array<uint8_t*, 3>, 4> ptrs{};
    for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 3; ++c)
            ptrs[a][c] = outs[a * 3 + c].data();
    ex::bc3::normalize_blocks_all_modes(b3, ptrs, 16);
    CHECK(std::memcmp(outs[3 * 3 + 1].data(), want3, 16) == 0 && std::memcmp(outs[0].data(), b3, 16) == 0);

    // common crate: colour arrays (SURVEY.md 8(c) hand-derived vector 0xF800 -> 0xBFD1 / 0x5FF1 / 0xBFE2)
    {
        using core::common::Color565;
        const uint16_t cols[4] = {0xF800, 0x07E0, 0x001F, 0xFFFF};
        uint16_t dec[4], rec[4];
        Color565::decorrelate_ycocg_r_ptr(cols, dec, 4, core::YCoCgVariant::Variant1);
        CHECK(dec[0] == 0xBFD1 && dec[1] == 0x783F && dec[2] == 0xF841 && dec[3] == 0xF820);
        Color565::recorrelate_ycocg_r_ptr(dec, rec, 4, core::YCoCgVariant::Variant1);
        CHECK(std::memcmp(rec, cols, 8) == 0);
        const uint16_t pairs[6] = {0x0100, 0x0302, 0x0504, 0x0706, 0x0908, 0x0B0A};
        uint16_t split[6], joined[6];
        core::common::split_color_endpoints(pairs, split, 12);
        const uint16_t want_split[6] = {0x0100, 0x0504, 0x0908, 0x0302, 0x0706, 0x0B0A};
        CHECK(std::memcmp(split, want_split, 12) == 0);
        Color565::recorrelate_ycocg_r_ptr_split(split, split + 3, joined, 6, core::YCoCgVariant::None);
        CHECK(std::memcmp(joined, pairs, 12) == 0);
    }
    // BC7 builder (additive): validation as the other builders, exact round trip
    {
        std::vector<uint8_t> b7 = gen_bc1(2000), t7(b7.size()), r7(b7.size());   // any bytes are valid BC7 input
        api::Bc7ManualTransformBuilder m7;
        CHECK(m7.transform(b7.data(), b7.size() - 1, t7.data(), t7.size()).kind == api::Error::InvalidLength);
        CHECK(m7.transform(b7.data(), b7.size(), t7.data(), 16).kind == api::Error::OutputBufferTooSmall);
        CHECK(m7.transform(b7.data(), b7.size(), t7.data(), t7.size()).is_ok());
        CHECK(m7.untransform(t7.data(), t7.size(), r7.data(), r7.size()).is_ok());
        CHECK(r7 == b7 && t7 != b7);
    }
    // util: decoders (bc1_decode.rs / bc3_decode.rs unit vectors)
    {
        namespace ut = core::util;
        const uint8_t red[8] = {0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0};
        const ut::Decoded4x4Block d = ut::decode_bc1_block(red);
        CHECK(d.has_identical_pixels() && d.get_pixel_unchecked(2, 3) == (ut::Color8888{255, 0, 0, 255}));
        const uint8_t bc3_block[16] = {0, 0, 0, 255, 255, 255, 255, 255, 255, 255, 18, 0, 0, 0, 0, 250};
        const ut::Decoded4x4Block e = ut::decode_bc3_block(bc3_block);
        CHECK(e.pixels[0] == (ut::Color8888{255, 255, 255, 0}) && e.pixels[3] == (ut::Color8888{255, 255, 255, 255}));
        CHECK(e.pixels[12] == (ut::Color8888{170, 170, 219, 255}) && e.pixels[15] == (ut::Color8888{85, 85, 183, 255}));
        std::vector<uint8_t> blocks = gen_bc1(1000), other = blocks;
        other[8 * 17 + 4] ^= 0x03;   // one index of block 17
        std::vector<ut::Decoded4x4Block> px(1000), px2(1000);
        ut::decode_bc1_blocks(blocks.data(), px.data(), 1000);
        ut::decode_bc1_blocks(other.data(), px2.data(), 1000);
        size_t differing = 0;
        for (size_t i = 0; i < 1000; ++i)
            differing += std::memcmp(&px[i], &px2[i], sizeof px[i]) != 0;
        CHECK(ut::count_pixel_differences(1, blocks.data(), other.data(), blocks.size()) == differing);
        CHECK(ut::count_pixel_differences(1, blocks.data(), blocks.data(), blocks.size()) == 0);
    }
}

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    try {
        cpu_tests();
        if (gpu) gpu_tests();
    } catch (const DeviceError& e) {
        std::printf("DeviceError %d: %s\n", e.code, e.what());
        return 2;
    }
    std::printf("%s: %d failure(s)\n", gpu ? "gpu" : "cpu", failures);
    return failures ? 1 : 0;
}
