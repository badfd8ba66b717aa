//! Replacement bodies for rofl_crypto/src/rand_proof_vec/mod.rs:14-118: d independent randomness proofs in one launch.
use curve25519_dalek_ng::ristretto::RistrettoPoint;
use curve25519_dalek_ng::scalar::Scalar;

use crate::ffi::*;
use crate::rand_proof::{ElGamalPair, RandProof, RandProofError};

const PROOF_LEN: usize = 128;   // RandProof::to_bytes (rand_proof/mod.rs:91-97)
const PAIR_LEN: usize = 64;     // ElGamalPair::to_bytes (rand_proof/el_gamal.rs:105-110)

fn run(value_vec: &Vec<f32>, existing: Option<&Vec<RistrettoPoint>>, random_vec: &Vec<Scalar>)
       -> Result<(Vec<RandProof>, Vec<ElGamalPair>), RandProofError> {
    if value_vec.len() != random_vec.len() { return Err(RandProofError::WrongNumBlindingFactors); }   // :18-20, :54-56
    let d = value_vec.len();
    let r = scalars_to_bytes(random_vec);
    let ex = existing.map(|v| points_to_bytes(v));
    let (mut proofs, mut pairs) = (vec![0u8; d * PROOF_LEN], vec![0u8; d * PAIR_LEN]);
    let nonce = fresh_nonce();
    let rc = unsafe {
        rofl_create_randproof_vec(value_vec.as_ptr(), d, r.as_ptr(), random_vec.len(), ex.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()),
                                  fp_bits(), fp_frac(), &nonce, proofs.as_mut_ptr(), pairs.as_mut_ptr())
    };
    if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }
    Ok((proofs.chunks(PROOF_LEN).map(|p| RandProof::from_bytes(p).unwrap()).collect(),
        pairs.chunks(PAIR_LEN).map(|p| ElGamalPair::from_bytes(p).unwrap()).collect()))
}

pub fn create_randproof_vec(value_vec: &Vec<f32>, random_vec: &Vec<Scalar>) -> Result<(Vec<RandProof>, Vec<ElGamalPair>), RandProofError> {
    run(value_vec, None, random_vec)
}
pub fn create_randproof_vec_existing(value_vec: &Vec<f32>, existing_value_com_vec: Vec<RistrettoPoint>, random_vec: &Vec<Scalar>)
    -> Result<(Vec<RandProof>, Vec<ElGamalPair>), RandProofError> {
    run(value_vec, Some(&existing_value_com_vec), random_vec)
}
pub fn verify_randproof_vec(randproof_vec: &Vec<RandProof>, commit_vec: &Vec<ElGamalPair>) -> Result<bool, RandProofError> {
    if randproof_vec.len() != commit_vec.len() { return Err(RandProofError::WrongNumberOfElGamalPairs); }   // :95-97
    let pb: Vec<u8> = randproof_vec.iter().flat_map(|p| p.to_bytes()).collect();
    let cb: Vec<u8> = commit_vec.iter().flat_map(|c| c.to_bytes()).collect();
    let mut ok: std::os::raw::c_int = 0;
    let rc = unsafe { rofl_verify_randproof_vec(pb.as_ptr(), cb.as_ptr(), randproof_vec.len(), &mut ok) };
    if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }
    Ok(ok != 0)
}
