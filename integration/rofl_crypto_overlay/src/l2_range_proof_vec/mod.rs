//! Replacement bodies for rofl_crypto/src/l2_range_proof_vec/mod.rs:15-253.
use bulletproofs::{ProofError, RangeProof};
use curve25519_dalek_ng::ristretto::{CompressedRistretto, RistrettoPoint};
use curve25519_dalek_ng::scalar::Scalar;

use crate::ffi::*;
pub mod errors;
pub use self::errors::L2RangeProofError;

/// reference: l2_range_proof_vec/mod.rs:15-140 (sum of squares mod l, f32 shadow check, label b"L2RangeProof", gens (64, 1))
pub fn create_rangeproof_l2(
    value_vec: &Vec<f32>,
    blinding_vec: &Vec<Scalar>,
    prove_range: usize,
    n_partition: usize,
) -> Result<(RangeProof, RistrettoPoint), L2RangeProofError> {
    let bl = scalars_to_bytes(blinding_vec);
    let mut proof = vec![0u8; 32 * (9 + 2 * 6)];         // n = 64, m = 1 is the largest shape
    let mut commit = [0u8; 32];
    let mut plen = 0usize;
    let nonce = fresh_nonce();
    let rc = unsafe {
        rofl_create_rangeproof_l2(value_vec.as_ptr(), value_vec.len(), bl.as_ptr(), blinding_vec.len(), prove_range, n_partition,
                                  fp_bits(), fp_frac(), &nonce, proof.as_mut_ptr(), &mut plen, commit.as_mut_ptr())
    };
    match rc {
        ROFL_OK => Ok((RangeProof::from_bytes(&proof[..plen]).expect("librofl_zk proof layout"),
                       CompressedRistretto(commit).decompress().expect("valid encoding"))),
        ROFL_WRONG_NUM_BLINDING_FACTORS => Err(ProofError::WrongNumBlindingFactors.into()),
        ROFL_NORM_OUT_OF_RANGE => Err(L2RangeProofError::NormOutOfRangeError(last_error())),   // :60-64
        ROFL_OVERFLOW => Err(L2RangeProofError::OverflowError(last_error(), String::new())),   // :53-58
        ROFL_INVALID_BITSIZE => Err(ProofError::InvalidBitsize.into()),
        ROFL_SUM_ERROR => Err(L2RangeProofError::SumError),
        _ => panic!("Should not get here: {}", last_error()),
    }
}

/// reference: l2_range_proof_vec/mod.rs:185-253
pub fn verify_rangeproof_l2(range_proof: &RangeProof, commit: &RistrettoPoint, prove_range: usize) -> Result<bool, ProofError> {
    let pb = range_proof.to_bytes();
    let cb = commit.compress().to_bytes();
    let seed = fresh_seed();
    let mut ok: std::os::raw::c_int = 0;
    let rc = unsafe { rofl_verify_rangeproof_l2(pb.as_ptr(), pb.len(), cb.as_ptr(), prove_range, fp_bits(), fp_frac(), seed.as_ptr(), &mut ok) };
    match rc {
        ROFL_OK => Ok(ok != 0),
        ROFL_INVALID_BITSIZE => Err(ProofError::InvalidBitsize),
        ROFL_FORMAT_ERROR => Err(ProofError::FormatError),
        _ => panic!("rofl_zk: {}", last_error()),
    }
}
