//! Replacement bodies for rofl_crypto/src/square_rand_proof_vec/mod.rs:18-159.
use curve25519_dalek_ng::ristretto::RistrettoPoint;
use curve25519_dalek_ng::scalar::Scalar;

use crate::ffi::*;
use crate::l2_range_proof_vec::errors::L2RangeProofError;
use crate::square_rand_proof::pedersen::SquareRandProofCommitments;
use crate::square_rand_proof::SquareRandProof;

const PROOF_LEN: usize = 192;   // SquareRandProof::to_bytes (square_rand_proof/mod.rs:118-125)
const COMMIT_LEN: usize = 96;   // SquareRandProofCommitments::to_bytes (square_rand_proof/pedersen.rs:25-30)

fn run(value_vec: &Vec<f32>, existing: Option<&Vec<RistrettoPoint>>, random_vec: &Vec<Scalar>, random_vec_2: &Vec<Scalar>)
       -> Result<(Vec<SquareRandProof>, Vec<SquareRandProofCommitments>), L2RangeProofError> {
    if value_vec.len() != random_vec.len() { return Err(L2RangeProofError::WrongNumBlindingFactors); }   // :24-26, :81-83
    let d = value_vec.len();
    let (r1, r2) = (scalars_to_bytes(random_vec), scalars_to_bytes(random_vec_2));
    let ex = existing.map(|v| points_to_bytes(v));
    let (mut proofs, mut commits) = (vec![0u8; d * PROOF_LEN], vec![0u8; d * COMMIT_LEN]);
    let nonce = fresh_nonce();
    let rc = unsafe {
        rofl_create_squarerandproof_vec(value_vec.as_ptr(), d, r1.as_ptr(), random_vec.len(), r2.as_ptr(),
                                        ex.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()), fp_bits(), fp_frac(), &nonce,
                                        proofs.as_mut_ptr(), commits.as_mut_ptr())
    };
    if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }
    Ok((proofs.chunks(PROOF_LEN).map(|p| SquareRandProof::from_bytes(p).unwrap()).collect(),
        commits.chunks(COMMIT_LEN).map(|c| SquareRandProofCommitments::from_bytes(c).unwrap()).collect()))
}

pub fn create_l2rangeproof_vec_existing(value_vec: &Vec<f32>, value_com_vec: Vec<RistrettoPoint>, random_vec: &Vec<Scalar>,
                                        random_vec_2: &Vec<Scalar>)
    -> Result<(Vec<SquareRandProof>, Vec<SquareRandProofCommitments>), L2RangeProofError> {
    run(value_vec, Some(&value_com_vec), random_vec, random_vec_2)
}
pub fn create_l2rangeproof_vec(value_vec: &Vec<f32>, random_vec: &Vec<Scalar>, random_vec_2: &Vec<Scalar>)
    -> Result<(Vec<SquareRandProof>, Vec<SquareRandProofCommitments>), L2RangeProofError> {
    run(value_vec, None, random_vec, random_vec_2)
}
pub fn verify_l2rangeproof_vec(randproof_vec: &Vec<SquareRandProof>, commit_vec: &Vec<SquareRandProofCommitments>)
    -> Result<bool, L2RangeProofError> {
    if randproof_vec.len() != commit_vec.len() { return Err(L2RangeProofError::WrongNumberOfElGamalPairs); }   // :133-135
    let pb: Vec<u8> = randproof_vec.iter().flat_map(|p| p.to_bytes()).collect();
    let cb: Vec<u8> = commit_vec.iter().flat_map(|c| c.to_bytes()).collect();
    let mut ok: std::os::raw::c_int = 0;
    let rc = unsafe { rofl_verify_squarerandproof_vec(pb.as_ptr(), cb.as_ptr(), randproof_vec.len(), &mut ok) };
    if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }
    Ok(ok != 0)
}
