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
