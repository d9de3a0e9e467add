//! `ldpc-toolbox ber` with the GPU decoders added to the reference's list, after the reference's
//! examples/external_decoder_ber.rs:  cargo run --release --example hip_ber -- code.alist \
//!     --decoder Hip-Minsumf32 --min-ebn0 1.0 --max-ebn0 2.0 --step-ebn0 0.25 --num-threads 2
//! (`BerTest` decodes one frame per worker per call, so this is the convenient route, not the fast one:
//! the throughput route is `HipDecoder::decode_batch` or `python -m ldpc_toolbox_amd.ber`.)
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

/// GPU implementations offered on the command line, `Hip-` + the library's name
const HIP_NAMES: &[&str] = &["Minsumf32", "HLMinsumf32", "Phif64", "Tanhf32", "HLTanhf32", "Aminstari8"];

#[derive(Debug, Clone, Copy, Eq, PartialEq, Hash)]
enum DecoderImplementation {
    Reference(factory::DecoderImplementation),
    Hip(&'static str),
}

impl DecoderFactory for DecoderImplementation {
    fn build_decoder(&self, h: SparseMatrix) -> Box<dyn LdpcDecoder> {
        match self {
            DecoderImplementation::Reference(d) => d.build_decoder(h),
            DecoderImplementation::Hip(name) => Box::new(HipDecoder::new(&h, name).expect("HIP decoder")),
        }
    }
}

impl Display for DecoderImplementation {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> Result<(), std::fmt::Error> {
        match self {
            DecoderImplementation::Reference(d) => d.fmt(f),
            DecoderImplementation::Hip(name) => write!(f, "Hip-{name}"),
        }
    }
}

impl ValueEnum for DecoderImplementation {
    fn value_variants<'a>() -> &'a [Self] {
        static VARIANTS: LazyLock<Vec<DecoderImplementation>> = LazyLock::new(|| {
            let mut v = factory::DecoderImplementation::value_variants()
                .iter()
                .map(|&d| DecoderImplementation::Reference(d))
                .collect::<Vec<_>>();
            v.extend(HIP_NAMES.iter().map(|&n| DecoderImplementation::Hip(n)));
            v
        });
        &VARIANTS
    }

    fn to_possible_value(&self) -> Option<PossibleValue> {
        match self {
            DecoderImplementation::Reference(d) => d.to_possible_value(),
            DecoderImplementation::Hip(name) => {
                Some(PossibleValue::new(format!("Hip-{name}")).help("MI355X decoder (libldpc_toolbox.so)"))
            }
        }
    }
}

#[termination::display]
fn main() -> Result<(), Box<dyn Error>> {
    Args::<DecoderImplementation>::parse().run()
}
