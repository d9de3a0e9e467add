//! The reference's `ber` command line driving the GPU decoders:
//!     cargo run --release --example hip_ber -- code.alist --decoder HLTanhf32 \
//!         --min-ebn0 1.0 --max-ebn0 2.0 --step-ebn0 0.25 --num-threads 2
//! Modelled on the reference's own plug-in example (examples/external_decoder_ber.rs:33-79): the decoder
//! enum EXTENDS `factory::DecoderImplementation` -- every one of its 36 names is offered and built on the GPU
//! under the same name -- by the four min-sum names this library adds (40 in all).
//! `BerTest` decodes one frame per worker per call, so this is the convenient route, not the fast one: the
//! throughput route is `HipDecoder::decode_batch` or `python -m ldpc_toolbox_amd.ber`.
//! (Not compiled in the build image: it has no Rust toolchain.)
use clap::{Parser, ValueEnum, builder::PossibleValue};
use ldpc_toolbox::{
    cli::{Run, ber::Args},
    decoder::{
        LdpcDecoder,
        factory::{self, DecoderFactory},
    },
    sparse::SparseMatrix,
};
use ldpc_toolbox_hip::HipDecoder;
use std::{error::Error, fmt::Display, sync::LazyLock};

/// The min-sum family of libldpc_toolbox.so (not in the reference: SURVEY.md F2 / Appendix A.6).
#[derive(Debug, Clone, Copy, Eq, PartialEq, Hash)]
#[allow(clippy::upper_case_acronyms)]
enum Minsum {
    Minsumf32,
    Minsumf64,
    HLMinsumf32,
    HLMinsumf64,
}

impl Minsum {
    const ALL: [Minsum; 4] = [Minsum::Minsumf32, Minsum::Minsumf64, Minsum::HLMinsumf32, Minsum::HLMinsumf64];
    fn name(self) -> &'static str {
        match self {
            Minsum::Minsumf32 => "Minsumf32",
            Minsum::Minsumf64 => "Minsumf64",
            Minsum::HLMinsumf32 => "HLMinsumf32",
            Minsum::HLMinsumf64 => "HLMinsumf64",
        }
    }
}

/// Extends ldpc_toolbox's `DecoderImplementation` the way the reference's example does: every variant of
/// the reference enum, then the additions.  All of them are built on the GPU.
#[derive(Debug, Clone, Copy, Eq, PartialEq, Hash)]
enum DecoderImplementation {
    /// one of the reference's 36 implementations (src/decoder/factory.rs:240-277), decoded by the HIP library
    DecoderImplementation(factory::DecoderImplementation),
    /// one of the four min-sum implementations the HIP library adds
    Minsum(Minsum),
}

impl DecoderImplementation {
    /// The name libldpc_toolbox.so understands: the reference's own `Display` of the variant (its names are
    /// verbatim, factory.rs:31-33 `rename_all = "verbatim"`), which is what the C ABI's `implementation`
    /// argument takes (src/c_api/decoder.rs:75-88 parses the same strings).
    fn library_name(&self) -> String {
        match self {
            DecoderImplementation::DecoderImplementation(d) => d.to_string(),
            DecoderImplementation::Minsum(m) => m.name().to_string(),
        }
    }
}

impl DecoderFactory for DecoderImplementation {
    fn build_decoder(&self, h: SparseMatrix) -> Box<dyn LdpcDecoder> {
        Box::new(HipDecoder::new(&h, &self.library_name()).expect("HIP decoder"))
    }
}

impl Display for DecoderImplementation {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> Result<(), std::fmt::Error> {
        match self {
            DecoderImplementation::DecoderImplementation(d) => d.fmt(f),
            DecoderImplementation::Minsum(m) => write!(f, "{}", m.name()),
        }
    }
}

impl ValueEnum for DecoderImplementation {
    fn value_variants<'a>() -> &'a [Self] {
        static VARIANTS: LazyLock<Vec<DecoderImplementation>> = LazyLock::new(|| {
            let mut variants = factory::DecoderImplementation::value_variants()
                .iter()
                .map(|&variant| DecoderImplementation::DecoderImplementation(variant))
                .collect::<Vec<_>>();
            variants.extend(Minsum::ALL.iter().map(|&m| DecoderImplementation::Minsum(m)));
            variants
        });
        &VARIANTS
    }

    fn to_possible_value(&self) -> Option<PossibleValue> {
        match self {
            DecoderImplementation::DecoderImplementation(a) => a.to_possible_value(),
            DecoderImplementation::Minsum(m) => {
                Some(PossibleValue::new(m.name()).help("min-sum (this library's addition), decoded on the GPU"))
            }
        }
    }
}

fn main() -> Result<(), Box<dyn Error>> {
    Args::<DecoderImplementation>::parse().run()
}
