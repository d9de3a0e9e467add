//! `LdpcDecoder` / `DecoderFactory` over the C ABI of libldpc_toolbox.so (include/ldpc_toolbox.h).
//!
//! The reference's traits (src/decoder.rs:19-35, src/decoder/factory.rs:19-25) decode one codeword
//! per call; `HipDecoder::decode` forwards to `ldpc_toolbox_decoder_decode_f64`.  A caller that owns
//! its frame loop should use `HipDecoder::decode_batch`, which decodes many codewords per call (the
//! GPU is idle most of the time with one frame in flight).

use ldpc_toolbox::decoder::{DecoderOutput, LdpcDecoder, factory::DecoderFactory};
use ldpc_toolbox::sparse::SparseMatrix;
use std::ffi::{CStr, CString, c_char, c_void};

#[link(name = "ldpc_toolbox")]
unsafe extern "C" {
    fn ldpc_toolbox_decoder_ctor_alist_string(
        alist: *const c_char,
        implementation: *const c_char,
        puncturing: *const c_char,
    ) -> *mut c_void;
    fn ldpc_toolbox_decoder_dtor(decoder: *mut c_void);
    fn ldpc_toolbox_decoder_decode_f64(
        decoder: *mut c_void,
        output: *mut u8,
        output_len: usize,
        llrs: *const f64,
        llrs_len: usize,
        max_iterations: u32,
    ) -> i32;
    fn ldpc_toolbox_decoder_decode_batch_f64(
        decoder: *mut c_void,
        output: *mut u8,
        output_len: usize,
        llrs: *const f64,
        llrs_len: usize,
        batch: usize,
        max_iterations: u32,
        iterations: *mut i32,
        posterior: *mut f64,
    ) -> i32;
    fn ldpc_toolbox_last_error() -> *const c_char;
}

fn last_error() -> String {
    let p = unsafe { ldpc_toolbox_last_error() };
    if p.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

/// One decoder handle on one GPU.  `&mut self` on every call: one call at a time per handle, as
/// the C ABI requires; distinct handles are independent.
#[derive(Debug)]
pub struct HipDecoder {
    handle: *mut c_void,
    n: usize,
}

// the handle owns device memory and a stream; nothing in it is tied to the creating thread
unsafe impl Send for HipDecoder {}

impl HipDecoder {
    /// `implementation`: any name of the reference (`Phif64`, `HLTanhf32`, `Aminstari8Jones`, ...) or
    /// this library's `Minsumf32/f64`, `HLMinsumf32/f64`; append `@hip:N` to pick GPU N.
    pub fn new(h: &SparseMatrix, implementation: &str) -> Result<HipDecoder, String> {
        let alist = CString::new(h.alist()).map_err(|e| e.to_string())?;
        let name = CString::new(implementation).map_err(|e| e.to_string())?;
        let handle =
            unsafe { ldpc_toolbox_decoder_ctor_alist_string(alist.as_ptr(), name.as_ptr(), c"".as_ptr()) };
        if handle.is_null() {
            return Err(format!("HIP decoder unavailable: {}", last_error()));
        }
        Ok(HipDecoder {
            handle,
            n: h.num_cols(),
        })
    }

    /// `llrs`: `batch` rows of `n` LLRs.  Returns one `Result` per codeword, in order.
    pub fn decode_batch(
        &mut self,
        llrs: &[f64],
        max_iterations: usize,
    ) -> Result<Vec<Result<DecoderOutput, DecoderOutput>>, String> {
        assert_eq!(llrs.len() % self.n, 0);
        let batch = llrs.len() / self.n;
        let mut bits = vec![0u8; batch * self.n];
        let mut iterations = vec![0i32; batch];
        let rc = unsafe {
            ldpc_toolbox_decoder_decode_batch_f64(
                self.handle,
                bits.as_mut_ptr(),
                self.n,
                llrs.as_ptr(),
                self.n,
                batch,
                max_iterations as u32,
                iterations.as_mut_ptr(),
                std::ptr::null_mut(),
            )
        };
        if rc != 0 {
            return Err(last_error());
        }
        Ok(bits
            .chunks_exact(self.n)
            .zip(iterations)
            .map(|(cw, it)| {
                let codeword = cw.to_vec();
                if it >= 0 {
                    Ok(DecoderOutput {
                        codeword,
                        iterations: it as usize,
                    })
                } else {
                    Err(DecoderOutput {
                        codeword,
                        iterations: max_iterations,
                    })
                }
            })
            .collect())
    }
}

impl LdpcDecoder for HipDecoder {
    fn decode(
        &mut self,
        llrs: &[f64],
        max_iterations: usize,
    ) -> Result<DecoderOutput, DecoderOutput> {
        assert_eq!(llrs.len(), self.n); // the reference's decoders assert the same (flooding.rs:56)
        let mut codeword = vec![0u8; self.n];
        let it = unsafe {
            ldpc_toolbox_decoder_decode_f64(
                self.handle,
                codeword.as_mut_ptr(),
                self.n,
                llrs.as_ptr(),
                llrs.len(),
                max_iterations as u32,
            )
        };
        if it >= 0 {
            Ok(DecoderOutput {
                codeword,
                iterations: it as usize,
            })
        } else if it == -1 {
            // the reference's Err(..): max_iterations used, output = last hard decision
            Err(DecoderOutput {
                codeword,
                iterations: max_iterations,
            })
        } else {
            // LDPC_TOOLBOX_ERR_* (include/ldpc_toolbox.h): the call itself failed (GPU fault, bad
            // length) and nothing was decoded.  Counting that as a decoding failure would corrupt
            // BER statistics silently -- abort like the reference's own panics do.
            panic!("ldpc_toolbox (hip): decode failed with error {}: {}", it, last_error());
        }
    }
}

impl Drop for HipDecoder {
    fn drop(&mut self) {
        unsafe { ldpc_toolbox_decoder_dtor(self.handle) }
    }
}

/// Factory for `BerTest` / the `ber` command line: every worker thread gets its own handle.
#[derive(Debug, Clone, PartialEq, Eq, Hash)]
pub struct HipFactory {
    /// implementation name handed to the library, e.g. `"Minsumf32"` or `"HLTanhf32@hip:1"`
    pub implementation: String,
}

impl std::fmt::Display for HipFactory {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "{}", self.implementation)
    }
}

impl DecoderFactory for HipFactory {
    fn build_decoder(&self, h: SparseMatrix) -> Box<dyn LdpcDecoder> {
        Box::new(HipDecoder::new(&h, &self.implementation).expect("HIP decoder"))
    }
}

/// The reference's own decoder known answers (ldpc-toolbox `src/decoder/flooding.rs`, `mod test`: example 2.5 / 2.23 of
/// S. J. Johnson, "Iterative Error Correction") through this shim: `cargo test` on a machine with an MI355X and
/// `libldpc_toolbox.so` built (INTEGRATION.md section 2 lists the commands and the expected lines).
#[cfg(test)]
mod tests {
    use super::*;

    fn h() -> SparseMatrix {
        let mut h = SparseMatrix::new(4, 6);
        h.insert_row(0, [0, 1, 3].iter());
        h.insert_row(1, [1, 2, 4].iter());
        h.insert_row(2, [0, 4, 5].iter());
        h.insert_row(3, [2, 3, 5].iter());
        h
    }

    fn to_llrs(bits: &[u8]) -> Vec<f64> {
        bits.iter().map(|&b| if b == 0 { 1.3863 } else { -1.3863 }).collect()
    }

    #[test]
    fn reference_kat_through_the_trait() {
        // every decoder the `ber` command line could ask for by this name goes through `DecoderFactory`
        let factory = HipFactory { implementation: "Phif64".to_string() };
        let mut decoder = factory.build_decoder(h());
        let good = [0u8, 0, 1, 0, 1, 1];
        let out = decoder.decode(&to_llrs(&good), 100).unwrap();
        assert_eq!(&out.codeword, &good);
        assert_eq!(out.iterations, 0);
        for j in 0..good.len() {
            let mut bad = good;
            bad[j] ^= 1;
            let out = decoder.decode(&to_llrs(&bad), 100).unwrap();
            assert_eq!(&out.codeword, &good);
            assert_eq!(out.iterations, 1);
        }
    }

    #[test]
    fn batch_call_equals_the_scalar_calls() {
        let mut decoder = HipDecoder::new(&h(), "Minsumf32").unwrap();
        let good = [0u8, 0, 1, 0, 1, 1];
        let mut rows = to_llrs(&good);
        for j in 0..good.len() {
            let mut bad = good;
            bad[j] ^= 1;
            rows.extend(to_llrs(&bad));
        }
        let batch = decoder.decode_batch(&rows, 100).unwrap();
        assert_eq!(batch.len(), 7);
        for (row, got) in rows.chunks_exact(6).zip(batch) {
            let want = decoder.decode(row, 100);
            assert_eq!(got.is_ok(), want.is_ok());
            let (g, w) = (got.unwrap_or_else(|e| e), want.unwrap_or_else(|e| e));
            assert_eq!(g.codeword, w.codeword);
            assert_eq!(g.iterations, w.iterations);
        }
    }
}
