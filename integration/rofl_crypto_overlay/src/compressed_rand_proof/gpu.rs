//! Replacement bodies for the three helpers of rofl_crypto/src/compressed_rand_proof/mod.rs:134-160 (`helper_prove`,
//! `helper_prove_existing`, `helper_verify`): delete them there and add `mod gpu;` -- a second `impl` block in a child module sees the
//! private fields.  ONE 128-byte proof for all d ElGamal pairs; the device computes the pairs and the challenge-power dot products, the
//! transcript over the d pairs (labels UNIQUE_U8_TRIPLETS[i], unique_u8_triplets.rs) stays a host sponge inside the library.
//! NOT compiled in the build image (no Rust toolchain); the C entry points are tested through ctypes (tests/test_gpu_parity.py).
use curve25519_dalek_ng::ristretto::RistrettoPoint;
use curve25519_dalek_ng::scalar::Scalar;

use super::types::CompressedRandProofCommitments;
use super::{CompressedRandProof, ElGamalPair, ProofError};
use crate::ffi::*;

const PAIR_LEN: usize = 64;     // ElGamalPair::to_bytes: L | R (rand_proof/el_gamal.rs)

fn prove(m_vec: &Vec<f32>, m_com: Option<&Vec<RistrettoPoint>>, r_vec: &Vec<Scalar>)
    -> Result<(CompressedRandProof, CompressedRandProofCommitments), ProofError> {
    let d = m_vec.len();
    if r_vec.len() != d || m_com.map_or(false, |c| c.len() != d) { return Err(ProofError::WrongNumBlindingFactors); }   // party.rs:60-62
    let r = scalars_to_bytes(r_vec);
    let ex = m_com.map(|v| points_to_bytes(v));
    let (mut proof, mut pairs) = (vec![0u8; CompressedRandProof::serialized_size()], vec![0u8; d * PAIR_LEN]);
    let nonce = fresh_nonce();
    let rc = unsafe {
        rofl_create_compressed_randproof(m_vec.as_ptr(), d, r.as_ptr(), r_vec.len(), ex.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()),
                                         fp_bits(), fp_frac(), &nonce, proof.as_mut_ptr(), pairs.as_mut_ptr())
    };
    if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }
    let c_vec: Vec<ElGamalPair> = pairs.chunks(PAIR_LEN).map(|c| ElGamalPair::from_bytes(c).unwrap()).collect();
    Ok((CompressedRandProof::from_bytes(&proof)?, CompressedRandProofCommitments { c_vec }))
}

impl CompressedRandProof {
    pub fn helper_prove(m_vec: &Vec<f32>, r_vec: Vec<Scalar>) -> Result<(CompressedRandProof, CompressedRandProofCommitments), ProofError> {
        prove(m_vec, None, &r_vec)
    }
    pub fn helper_prove_existing(m_vec: &Vec<f32>, m_com: Vec<RistrettoPoint>, r_vec: Vec<Scalar>)
        -> Result<(CompressedRandProof, CompressedRandProofCommitments), ProofError> {
        prove(m_vec, Some(&m_com), &r_vec)
    }
    pub fn helper_verify(&self, c_vec: Vec<ElGamalPair>) -> Result<(), ProofError> {
        let pairs: Vec<u8> = c_vec.iter().flat_map(|c| c.to_bytes()).collect();
        let proof = self.to_bytes();
        let mut ok: std::os::raw::c_int = 0;
        let rc = unsafe { rofl_verify_compressed_randproof(proof.as_ptr(), pairs.as_ptr(), c_vec.len(), &mut ok) };
        if rc == ROFL_FORMAT_ERROR { return Err(ProofError::FormatError); }
        if rc != ROFL_OK { panic!("rofl_zk: {}", last_error()); }
        if ok != 0 { Ok(()) } else { Err(ProofError::VerificationError) }
    }
}
