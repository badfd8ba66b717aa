//! GPU bodies for the vector operations of rofl_crypto/src/pedersen_ops.rs:9-59, 103-108 (same signatures; the scalar-only
//! helpers of that file -- add_scalar_vec, zero_*_vec, rnd_scalar_vec, generate_cancelling_scalar_vec -- stay as they are).
use curve25519_dalek_ng::ristretto::RistrettoPoint;
use curve25519_dalek_ng::scalar::Scalar;

use crate::ffi::*;
use crate::fp::BSGS_N_BITS;

pub fn commit_no_blinding_vec(scalar_vec: &Vec<Scalar>) -> Vec<RistrettoPoint> {
    let v = scalars_to_bytes(scalar_vec);
    let mut out = vec![0u8; scalar_vec.len() * 32];
    let rc = unsafe { rofl_commit_vec(v.as_ptr(), std::ptr::null(), scalar_vec.len(), out.as_mut_ptr()) };
    assert!(rc == ROFL_OK, "rofl_zk: {}", last_error());
    bytes_to_points(&out)
}
pub fn commit_vec(scalar_vec: &Vec<Scalar>, blinding_vec: &Vec<Scalar>) -> Vec<RistrettoPoint> {
    let (v, b) = (scalars_to_bytes(scalar_vec), scalars_to_bytes(blinding_vec));
    let mut out = vec![0u8; scalar_vec.len() * 32];
    let rc = unsafe { rofl_commit_vec(v.as_ptr(), b.as_ptr(), scalar_vec.len(), out.as_mut_ptr()) };
    assert!(rc == ROFL_OK, "rofl_zk: {}", last_error());
    bytes_to_points(&out)
}
pub fn add_rp_vec(a: &Vec<RistrettoPoint>, b: &Vec<RistrettoPoint>) -> Vec<RistrettoPoint> {
    let (x, y) = (points_to_bytes(a), points_to_bytes(b));
    let mut out = vec![0u8; a.len() * 32];
    let rc = unsafe { rofl_add_points_vec(x.as_ptr(), y.as_ptr(), a.len(), out.as_mut_ptr()) };
    assert!(rc == ROFL_OK, "rofl_zk: {}", last_error());
    bytes_to_points(&out)
}
pub fn compute_shifted_values_rp(rp_vec: &Vec<RistrettoPoint>, offset: &RistrettoPoint) -> Vec<RistrettoPoint> {
    let x = points_to_bytes(rp_vec);
    let off = offset.compress().to_bytes();
    let mut out = vec![0u8; rp_vec.len() * 32];
    let rc = unsafe { rofl_shift_points(x.as_ptr(), rp_vec.len(), off.as_ptr(), out.as_mut_ptr()) };
    assert!(rc == ROFL_OK, "rofl_zk: {}", last_error());
    bytes_to_points(&out)
}
/// `BSGSTable` shrinks to its size: the baby-step table lives (cached) on the device.
pub fn discrete_log_vec(rp_vec: &Vec<RistrettoPoint>, table_size: usize) -> Vec<Scalar> {
    let x = points_to_bytes(rp_vec);
    let mut out = vec![0u8; rp_vec.len() * 32];
    let rc = unsafe { rofl_discrete_log_vec(x.as_ptr(), rp_vec.len(), table_size, BSGS_N_BITS as u32, out.as_mut_ptr()) };
    assert!(rc == ROFL_OK, "rofl_zk: {}", last_error());
    bytes_to_scalars(&out)
}

/// `enc_values.iter().map(|x| x.c_sq).sum()` of rofl_service/src/flserver/params.rs:220, 277 on the device: the sum of d compressed points
/// read every `stride` bytes (96 over serialized SquareRandProofCommitments, 32 over a packed vector); an empty input gives the identity.
pub fn sum_points_strided(bytes: &[u8], d: usize, stride: usize) -> RistrettoPoint {
    assert!(stride >= 32 && bytes.len() >= d.saturating_sub(1) * stride + if d > 0 { 32 } else { 0 });
    let mut out = [0u8; 32];
    let rc = unsafe { rofl_sum_points(bytes.as_ptr(), d, stride, out.as_mut_ptr()) };
    assert!(rc == ROFL_OK, "rofl_zk: {}", last_error());
    bytes_to_points(&out)[0]
}
pub fn sum_rp_vec(rp_vec: &Vec<RistrettoPoint>) -> RistrettoPoint { sum_points_strided(&points_to_bytes(rp_vec), rp_vec.len(), 32) }
