//! reference_kit, second program -- the CHOICES the reference's `transform_bcN_auto` makes, so that this repository's auto transform
//! (candidate order, the sections shown to the estimator, the strict `<`, the final re-transform) can be pinned too.
//!
//! Same rules as main.rs: not part of the product, never built in the repository's own image, std only plus the reference's crates by
//! path.  It calls
//!   transform_bc1_auto   core/dxt-lossless-transform-bc1/src/transform/transform_auto.rs:200
//!   transform_bc2_auto   core/dxt-lossless-transform-bc2/src/transform/transform_auto.rs:196
//!   transform_bc3_auto   core/dxt-lossless-transform-bc3/src/transform/transform_auto.rs:196
//! with `use_all_decorrelation_modes` false and true and ONE deterministic estimator that both sides can state in a line:
//!   size = CRC-32 of the section & 0xFFFFF, max_compressed_size = 0 (no scratch buffer)
//! -- every byte the estimator is shown matters, so a different candidate order, section or tie-break changes the choice.
//!
//!   auto <in_dir> <out_dir>
//! For every `<case>.<fmt>.in` it appends `<case>.<fmt>.all<0|1> <settings id as in main.rs> <length> <crc32 of the output>` to
//! `<out_dir>/AUTO_MANIFEST.txt`.
use std::{env, fs, io::Write, path::Path};

use dxt_lossless_transform_api_common::estimate::SizeEstimationOperations;
use dxt_lossless_transform_bc1::{transform_bc1_auto, Bc1EstimateSettings};
use dxt_lossless_transform_bc2::{transform_bc2_auto, Bc2EstimateSettings};
use dxt_lossless_transform_bc3::{transform_bc3_auto, Bc3EstimateSettings};

/// CRC-32 (IEEE 802.3, reflected, the one zlib.crc32 computes).
fn crc32(data: &[u8]) -> u32 {
    let mut crc = !0u32;
    for &b in data {
        crc ^= b as u32;
        for _ in 0..8 {
            crc = (crc >> 1) ^ (0xEDB8_8320 & (crc & 1).wrapping_neg());
        }
    }
    !crc
}

struct CrcEstimator;

impl SizeEstimationOperations for CrcEstimator {
    type Error = ();

    fn max_compressed_size(&self, _len_bytes: usize) -> Result<usize, Self::Error> {
        Ok(0)
    }

    unsafe fn estimate_compressed_size(
        &self,
        input_ptr: *const u8,
        len_bytes: usize,
        _output_ptr: *mut u8,
        _output_len: usize,
    ) -> Result<usize, Self::Error> {
        let section: &[u8] = if len_bytes == 0 { &[] } else { unsafe { std::slice::from_raw_parts(input_ptr, len_bytes) } };
        Ok((crc32(section) & 0x000F_FFFF) as usize)
    }
}

fn main() {
    let args: Vec<String> = env::args().collect();
    assert!(args.len() == 3, "usage: auto <in_dir> <out_dir>");
    let (in_dir, out_dir) = (Path::new(&args[1]), Path::new(&args[2]));
    fs::create_dir_all(out_dir).unwrap();
    let mut names: Vec<String> = fs::read_dir(in_dir).unwrap().map(|e| e.unwrap().file_name().into_string().unwrap()).collect();
    names.sort();
    let mut lines = Vec::new();
    for file in names.iter().filter(|n| n.ends_with(".in")) {
        let stem = file.strip_suffix(".in").unwrap();
        let (case, fmt) = stem.rsplit_once('.').expect("<case>.<fmt>.in");
        let input = fs::read(in_dir.join(file)).unwrap();
        let mut out = vec![0u8; input.len()];
        let len = input.len();
        for use_all in [false, true] {
            let id = match fmt {
                "bc1" => {
                    let options = Bc1EstimateSettings { size_estimator: CrcEstimator, use_all_decorrelation_modes: use_all };
                    let s = unsafe { transform_bc1_auto(input.as_ptr(), out.as_mut_ptr(), len, &options) }.expect("transform_bc1_auto");
                    format!("v{}c{}", s.decorrelation_mode as u8, s.split_colour_endpoints as u8)
                }
                "bc2" => {
                    let options = Bc2EstimateSettings { size_estimator: CrcEstimator, use_all_decorrelation_modes: use_all };
                    let s = unsafe { transform_bc2_auto(input.as_ptr(), out.as_mut_ptr(), len, &options) }.expect("transform_bc2_auto");
                    format!("v{}c{}", s.decorrelation_mode as u8, s.split_colour_endpoints as u8)
                }
                "bc3" => {
                    let options = Bc3EstimateSettings { size_estimator: CrcEstimator, use_all_decorrelation_modes: use_all };
                    let s = unsafe { transform_bc3_auto(input.as_ptr(), out.as_mut_ptr(), len, &options) }.expect("transform_bc3_auto");
                    format!("v{}a{}c{}", s.decorrelation_mode as u8, s.split_alpha_endpoints as u8, s.split_colour_endpoints as u8)
                }
                other => panic!("{file}: unknown format {other}"),
            };
            lines.push(format!("{case}.{fmt}.all{} {id} {} {:08x}", use_all as u8, out.len(), crc32(&out)));
        }
    }
    let mut f = fs::File::create(out_dir.join("AUTO_MANIFEST.txt")).unwrap();
    for line in &lines {
        writeln!(f, "{line}").unwrap();
    }
    eprintln!("{} auto transforms -> {}", lines.len(), out_dir.display());
}
