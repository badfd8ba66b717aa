//! Replacement bodies for rofl_crypto/src/square_proof_vec/mod.rs:18-159 (the square proofs over Pedersen commitments only: what the
//! `*Compressed` enc types of every paper e2e run use, ansible/experiments/cifar_large.yml:56,74,100).  Same signatures and errors.
//! NOT compiled in the build image (no Rust toolchain); the C entry points underneath are tested through ctypes
//! (tests/test_gpu_parity.py::test_square_proof_bit_exact, tests/test_gpu_l2_batch.py::test_l2_compressed_round).
use curve25519_dalek_ng::ristretto::RistrettoPoint;
use curve25519_dalek_ng::scalar::Scalar;

use crate::ffi::*;
use crate::square_proof::pedersen::SquareProofCommitments;
use crate::square_proof::SquareProof;
pub mod errors;
pub use self::errors::L2RangeProofError;

const PROOF_LEN: usize = 160;   // SquareProof::to_bytes: c_l' | c_sq' | Z_m | Z_r1 | Z_r2 (square_proof/mod.rs:118-125)
const COMMIT_LEN: usize = 64;   // SquareProofCommitments::to_bytes: c_l | c_sq (square_proof/pedersen.rs:23-28)

fn run(value_vec: &Vec<f32>, existing: Option<&Vec<RistrettoPoint>>, random_vec: &Vec<Scalar>, random_vec_2: &Vec<Scalar>)
       -> Result<(Vec<SquareProof>, Vec<SquareProofCommitments>), L2RangeProofError> {
    if value_vec.len() != random_vec.len() { return Err(L2RangeProofError::WrongNumBlindingFactors); }   // :24-26, :81-83
    assert_eq!(random_vec.len(), random_vec_2.len());                                                     // zip_eq panics (:95)
    let d = value_vec.len();
    let (r1, r2) = (scalars_to_bytes(random_vec), scalars_to_bytes(random_vec_2));
    let ex = existing.map(|v| points_to_bytes(v));
    let (mut proofs, mut commits) = (vec![0u8; d * PROOF_LEN], vec![0u8; d * COMMIT_LEN]);
    let nonce = fresh_nonce();
    let rc = unsafe {
        rofl_create_squareproof_vec(value_vec.as_ptr(), d, r1.as_ptr(), random_vec.len(), r2.as_ptr(),
                                    ex.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()), fp_bits(), fp_frac(), &nonce,
                                    proofs.as_mut_ptr(), commits.as_mut_ptr())
    };
    if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }
    Ok((proofs.chunks(PROOF_LEN).map(|p| SquareProof::from_bytes(p).unwrap()).collect(),
        commits.chunks(COMMIT_LEN).map(|c| SquareProofCommitments::from_bytes(c).unwrap()).collect()))
}

pub fn create_l2rangeproof_vec_existing(value_vec: &Vec<f32>, value_com_vec: Vec<RistrettoPoint>, random_vec: &Vec<Scalar>,
                                        random_vec_2: &Vec<Scalar>)
    -> Result<(Vec<SquareProof>, Vec<SquareProofCommitments>), L2RangeProofError> {
    run(value_vec, Some(&value_com_vec), random_vec, random_vec_2)
}
pub fn create_l2rangeproof_vec(value_vec: &Vec<f32>, random_vec: &Vec<Scalar>, random_vec_2: &Vec<Scalar>)
    -> Result<(Vec<SquareProof>, Vec<SquareProofCommitments>), L2RangeProofError> {
    run(value_vec, None, random_vec, random_vec_2)
}
pub fn verify_l2rangeproof_vec(randproof_vec: &Vec<SquareProof>, commit_vec: &Vec<SquareProofCommitments>)
    -> Result<bool, L2RangeProofError> {
    if randproof_vec.len() != commit_vec.len() { return Err(L2RangeProofError::WrongNumberOfElGamalPairs); }   // :133-135
    let pb: Vec<u8> = randproof_vec.iter().flat_map(|p| p.to_bytes()).collect();
    let cb: Vec<u8> = commit_vec.iter().flat_map(|c| c.to_bytes()).collect();
    let mut ok: std::os::raw::c_int = 0;
    let rc = unsafe { rofl_verify_squareproof_vec(pb.as_ptr(), cb.as_ptr(), randproof_vec.len(), &mut ok) };
    if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }      // (typed values cannot be malformed: from_bytes checked them)
    Ok(ok != 0)
}
/// Server side, all clients of a round in one launch sequence (params.rs:262-266 per client; server.rs:656-687): one verdict per client
/// and, as a by-product of decoding, every client's sum of c_sq (params.rs:277) for its L2 sum proof.
pub fn verify_l2rangeproof_vec_batch(proofs: &[&Vec<SquareProof>], commits: &[&Vec<SquareProofCommitments>])
    -> Result<(Vec<bool>, Vec<RistrettoPoint>), L2RangeProofError> {
    if proofs.len() != commits.len() { return Err(L2RangeProofError::WrongNumberOfElGamalPairs); }
    let n = proofs.len();
    if n == 0 { return Ok((vec![], vec![])); }
    let d = proofs[0].len();
    if proofs.iter().any(|p| p.len() != d) || commits.iter().any(|c| c.len() != d) { return Err(L2RangeProofError::WrongNumberOfElGamalPairs); }
    let pb: Vec<Vec<u8>> = proofs.iter().map(|v| v.iter().flat_map(|p| p.to_bytes()).collect()).collect();
    let cb: Vec<Vec<u8>> = commits.iter().map(|v| v.iter().flat_map(|c| c.to_bytes()).collect()).collect();
    let pp: Vec<*const u8> = pb.iter().map(|v| v.as_ptr()).collect();
    let cp: Vec<*const u8> = cb.iter().map(|v| v.as_ptr()).collect();
    let mut ok = vec![0 as std::os::raw::c_int; n];
    let mut sums = vec![0u8; 32 * n];
    let rc = unsafe { rofl_verify_squareproof_vec_batch(n, pp.as_ptr(), cp.as_ptr(), d, ok.as_mut_ptr(), sums.as_mut_ptr()) };
    if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }
    Ok((ok.iter().map(|&x| x != 0).collect(), bytes_to_points(&sums)))
}
