//! The reference's `ber` command line driving the GPU decoders:
//!     cargo run --release --example hip_ber -- code.alist --decoder Minsumf32 \
//!         --min-ebn0 1.0 --max-ebn0 2.0 --step-ebn0 0.25 --num-threads 2
//! `BerTest` decodes one frame per worker per call, so this is the convenient route, not the fast
//! one: the throughput route is `HipDecoder::decode_batch` or `python -m ldpc_toolbox_amd.ber`.
use clap::{Parser, ValueEnum};
use ldpc_toolbox::cli::{Run, ber::Args};
use ldpc_toolbox::decoder::{LdpcDecoder, factory::DecoderFactory};
use ldpc_toolbox::sparse::SparseMatrix;
use ldpc_toolbox_hip::HipDecoder;

/// The decoders offered by this command line.  Names are taken verbatim, as the reference's
/// `DecoderImplementation` does (src/decoder/factory.rs:31-33: `rename_all = "verbatim"`), so
/// `--decoder HLTanhf32` here means what it means to `ldpc-toolbox ber`.
#[derive(Debug, Clone, Copy, PartialEq, Eq, Hash, ValueEnum)]
#[value(rename_all = "verbatim")]
#[allow(clippy::upper_case_acronyms)]
enum Gpu {
    Minsumf32,
    HLMinsumf32,
    Phif64,
    Tanhf32,
    HLTanhf32,
    Aminstari8,
}

impl Gpu {
    /// name understood by libldpc_toolbox.so (src/decoder/factory.rs:240-277 plus the Minsum family):
    /// the variant name itself
    fn library_name(self) -> &'static str {
        match self {
            Gpu::Minsumf32 => "Minsumf32",
            Gpu::HLMinsumf32 => "HLMinsumf32",
            Gpu::Phif64 => "Phif64",
            Gpu::Tanhf32 => "Tanhf32",
            Gpu::HLTanhf32 => "HLTanhf32",
            Gpu::Aminstari8 => "Aminstari8",
        }
    }
}

impl std::fmt::Display for Gpu {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str(self.library_name())
    }
}

impl DecoderFactory for Gpu {
    fn build_decoder(&self, h: SparseMatrix) -> Box<dyn LdpcDecoder> {
        Box::new(HipDecoder::new(&h, self.library_name()).expect("HIP decoder"))
    }
}

fn main() -> Result<(), Box<dyn std::error::Error>> {
    Args::<Gpu>::parse().run()
}
