//! Replacement bodies for rofl_crypto/src/range_proof_vec/mod.rs:16-216 (same signatures, errors and panics).
use bulletproofs::{ProofError, RangeProof};
use curve25519_dalek_ng::ristretto::RistrettoPoint;
use curve25519_dalek_ng::scalar::Scalar;

use crate::ffi::*;
pub mod errors;
pub use self::errors::RangeProofError;

/// reference: range_proof_vec/mod.rs:16-102
pub fn create_rangeproof(
    value_vec: &Vec<f32>,
    blinding_vec: &Vec<Scalar>,
    prove_range: usize,
    n_partition: usize,
) -> Result<(Vec<RangeProof>, Vec<RistrettoPoint>), RangeProofError> {
    let d = value_vec.len();
    let bl = scalars_to_bytes(blinding_vec);
    let (n_proofs, plen) = unsafe { (rofl_rangeproof_chunks(d, n_partition), rofl_rangeproof_size(prove_range, d, n_partition)) };
    let mut proofs = vec![0u8; n_proofs * plen];
    let mut commits = vec![0u8; d * 32];
    let (mut plen_out, mut n_out) = (0usize, 0usize);
    let nonce = fresh_nonce();
    let rc = unsafe {
        rofl_create_rangeproof(value_vec.as_ptr(), d, bl.as_ptr(), blinding_vec.len(), prove_range, n_partition, fp_bits(), fp_frac(),
                               &nonce, proofs.as_mut_ptr(), &mut plen_out, &mut n_out, commits.as_mut_ptr())
    };
    match rc {
        ROFL_OK => Ok((
            proofs[..n_out * plen_out].chunks(plen_out).map(|p| RangeProof::from_bytes(p).expect("librofl_zk proof layout")).collect(),
            bytes_to_points(&commits),
        )),
        ROFL_WRONG_NUM_BLINDING_FACTORS => Err(ProofError::WrongNumBlindingFactors.into()),     // mod.rs:22-24
        ROFL_VALUE_OUT_OF_RANGE => Err(RangeProofError::ValueOutOfRangeError),                  // mod.rs:27-29
        ROFL_INVALID_BITSIZE => Err(ProofError::InvalidBitsize.into()),
        _ => panic!("Should not get here: {}", last_error()),                                  // mod.rs:137-140 panics likewise
    }
}

/// reference: range_proof_vec/mod.rs:149-191.  `Ok(false)` for a proof that does not verify, `Err` for malformed input.
pub fn verify_rangeproof(
    range_proof_vec: &Vec<RangeProof>,
    commit_vec: &Vec<RistrettoPoint>,
    prove_range: usize,
) -> Result<bool, ProofError> {
    let pb: Vec<u8> = range_proof_vec.iter().flat_map(|p| p.to_bytes()).collect();
    let cb = points_to_bytes(commit_vec);
    let seed = fresh_seed();
    let mut ok: std::os::raw::c_int = 0;
    let rc = unsafe {
        rofl_verify_rangeproof(pb.as_ptr(), pb.len() / range_proof_vec.len().max(1), range_proof_vec.len(), cb.as_ptr(), commit_vec.len(),
                               prove_range, fp_bits(), fp_frac(), seed.as_ptr(), &mut ok)
    };
    match rc {
        ROFL_OK => Ok(ok != 0),
        ROFL_INVALID_BITSIZE => Err(ProofError::InvalidBitsize),
        ROFL_INVALID_AGGREGATION => Err(ProofError::InvalidAggregation),
        ROFL_FORMAT_ERROR => Err(ProofError::FormatError),
        ROFL_INVALID_GENERATORS_LENGTH => Err(ProofError::InvalidGeneratorsLength),
        _ => panic!("rofl_zk: {}", last_error()),
    }
}

/// Not in the reference: all clients of a round in one call (server.rs:656-687 runs one pool task per client instead).
pub fn verify_rangeproof_batch(
    clients: &[(&Vec<RangeProof>, &Vec<RistrettoPoint>)],
    prove_range: usize,
) -> Result<Vec<bool>, ProofError> {
    if clients.is_empty() { return Ok(vec![]); }
    let pbs: Vec<Vec<u8>> = clients.iter().map(|(p, _)| p.iter().flat_map(|x| x.to_bytes()).collect()).collect();
    let cbs: Vec<Vec<u8>> = clients.iter().map(|(_, c)| points_to_bytes(c)).collect();
    let pp: Vec<*const u8> = pbs.iter().map(|v| v.as_ptr()).collect();
    let cp: Vec<*const u8> = cbs.iter().map(|v| v.as_ptr()).collect();
    let (n_proofs, d) = (clients[0].0.len(), clients[0].1.len());
    let seed = fresh_seed();
    let mut ok = vec![0 as std::os::raw::c_int; clients.len()];
    let rc = unsafe {
        rofl_verify_rangeproof_batch(clients.len(), pp.as_ptr(), pbs[0].len() / n_proofs.max(1), n_proofs, cp.as_ptr(), d, prove_range,
                                     fp_bits(), fp_frac(), seed.as_ptr(), ok.as_mut_ptr())
    };
    match rc { ROFL_OK => Ok(ok.iter().map(|&b| b != 0).collect()), ROFL_FORMAT_ERROR => Err(ProofError::FormatError), _ => panic!("rofl_zk: {}", last_error()) }
}
