//! reference_kit -- writes the bytes the REFERENCE produces, so that this repository's oracle and HIP path can be pinned to them.
//!
//! Not part of the product and not built in the repository's own image (it has no Rust toolchain): run it wherever `cargo` and a
//! checkout of Sewer56/dxt-lossless-transform exist, through `run.sh` next to this file.  It links the reference's three core
//! crates by path and calls exactly the functions the hot path is defined by:
//!   transform_bc1_with_settings   core/dxt-lossless-transform-bc1/src/transform/transform_with_settings.rs:31
//!   transform_bc2_with_settings   core/dxt-lossless-transform-bc2/src/transform/transform_with_settings.rs:30
//!   transform_bc3_with_settings   core/dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs:32
//! for every settings combination of `all_combinations()` (bc1 settings.rs:68, bc2 settings.rs, bc3 settings.rs:74).
//!
//!   reference_kit <in_dir> <out_dir>
//! For every `<case>.<fmt>.in` in <in_dir> (fmt = bc1 | bc2 | bc3; written by tests/golden/make_reference_inputs.py) it writes
//! `<case>.<fmt>.v<V>c<C>.out` (BC3: `v<V>a<A>c<C>`), V = the core `YCoCgVariant` number (None=0 … Variant3=3), A / C = split alpha /
//! colour endpoints, and one line `<file> <length> <crc32 hex>` per output into `<out_dir>/MANIFEST.txt`.  The untransform of every
//! output is checked against the input before anything is written.  std only.
use std::{env, fs, io::Write, path::Path};

use dxt_lossless_transform_bc1::{transform_bc1_with_settings, untransform_bc1_with_settings, Bc1TransformSettings};
use dxt_lossless_transform_bc2::{transform_bc2_with_settings, untransform_bc2_with_settings, Bc2TransformSettings};
use dxt_lossless_transform_bc3::{transform_bc3_with_settings, untransform_bc3_with_settings, Bc3TransformSettings};

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

fn emit(out_dir: &Path, manifest: &mut Vec<String>, name: String, input: &[u8], out: &[u8], back: &[u8]) {
    assert!(back == input, "{name}: the reference's untransform did not restore the input");
    fs::write(out_dir.join(&name), out).unwrap();
    manifest.push(format!("{name} {} {:08x}", out.len(), crc32(out)));
}

fn main() {
    let args: Vec<String> = env::args().collect();
    assert!(args.len() == 3, "usage: reference_kit <in_dir> <out_dir>");
    let (in_dir, out_dir) = (Path::new(&args[1]), Path::new(&args[2]));
    fs::create_dir_all(out_dir).unwrap();
    let mut names: Vec<String> = fs::read_dir(in_dir).unwrap().map(|e| e.unwrap().file_name().into_string().unwrap()).collect();
    names.sort();
    let mut manifest = Vec::new();
    for file in names.iter().filter(|n| n.ends_with(".in")) {
        let stem = file.strip_suffix(".in").unwrap();
        let (case, fmt) = stem.rsplit_once('.').expect("<case>.<fmt>.in");
        let input = fs::read(in_dir.join(file)).unwrap();
        let (mut out, mut back) = (vec![0u8; input.len()], vec![0u8; input.len()]);
        let len = input.len();
        match fmt {
            "bc1" => for s in Bc1TransformSettings::all_combinations() {
                assert!(len % 8 == 0);
                unsafe {
                    transform_bc1_with_settings(input.as_ptr(), out.as_mut_ptr(), len, s);
                    untransform_bc1_with_settings(out.as_ptr(), back.as_mut_ptr(), len, s);
                }
                let id = format!("v{}c{}", s.decorrelation_mode as u8, s.split_colour_endpoints as u8);
                emit(out_dir, &mut manifest, format!("{case}.bc1.{id}.out"), &input, &out, &back);
            },
            "bc2" => for s in Bc2TransformSettings::all_combinations() {
                assert!(len % 16 == 0);
                unsafe {
                    transform_bc2_with_settings(input.as_ptr(), out.as_mut_ptr(), len, s);
                    untransform_bc2_with_settings(out.as_ptr(), back.as_mut_ptr(), len, s);
                }
                let id = format!("v{}c{}", s.decorrelation_mode as u8, s.split_colour_endpoints as u8);
                emit(out_dir, &mut manifest, format!("{case}.bc2.{id}.out"), &input, &out, &back);
            },
            "bc3" => for s in Bc3TransformSettings::all_combinations() {
                assert!(len % 16 == 0);
                unsafe {
                    transform_bc3_with_settings(input.as_ptr(), out.as_mut_ptr(), len, s);
                    untransform_bc3_with_settings(out.as_ptr(), back.as_mut_ptr(), len, s);
                }
                let id = format!("v{}a{}c{}", s.decorrelation_mode as u8, s.split_alpha_endpoints as u8, s.split_colour_endpoints as u8);
                emit(out_dir, &mut manifest, format!("{case}.bc3.{id}.out"), &input, &out, &back);
            },
            other => panic!("{file}: unknown format {other}"),
        }
    }
    let mut f = fs::File::create(out_dir.join("MANIFEST.txt")).unwrap();
    for line in &manifest {
        writeln!(f, "{line}").unwrap();
    }
    eprintln!("{} outputs -> {}", manifest.len(), out_dir.display());
}
