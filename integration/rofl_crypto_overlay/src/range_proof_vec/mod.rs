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
    ensure_options();
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
/// The C entry point takes ONE (n_proofs, proof_len, d) for the whole batch and reads that many bytes from every client's
/// buffers -- and all three are chosen by the clients (params.rs:513-541 deserialises whatever arrives).  So the clients are
/// grouped by shape: the largest group goes through the batch entry, every other client through its own `verify_rangeproof`
/// call (a malformed set counts as "not verified"), and nobody is ever read with somebody else's lengths.
pub fn verify_rangeproof_batch(
    clients: &[(&Vec<RangeProof>, &Vec<RistrettoPoint>)],
    prove_range: usize,
) -> Result<Vec<bool>, ProofError> {
    if clients.is_empty() { return Ok(vec![]); }
    let pbs: Vec<Vec<u8>> = clients.iter().map(|(p, _)| p.iter().flat_map(|x| x.to_bytes()).collect()).collect();
    let cbs: Vec<Vec<u8>> = clients.iter().map(|(_, c)| points_to_bytes(c)).collect();
    // shape of a client: (number of proofs, bytes per proof, number of commitments)
    let shape = |i: usize| -> Option<(usize, usize, usize)> {
        let (np, d) = (clients[i].0.len(), clients[i].1.len());
        if np == 0 || d == 0 || pbs[i].len() % np != 0 { return None; }
        Some((np, pbs[i].len() / np, d))
    };
    let shapes: Vec<Option<(usize, usize, usize)>> = (0..clients.len()).map(shape).collect();
    let mut major: Option<(usize, usize, usize)> = None; let mut best = 0;
    for s in shapes.iter().flatten() {
        let n = shapes.iter().filter(|t| **t == Some(*s)).count();
        if n > best { best = n; major = Some(*s); }
    }
    let mut res = vec![false; clients.len()];
    let (np, plen, d) = match major { Some(m) => m, None => return Ok(res) };
    let idx: Vec<usize> = (0..clients.len()).filter(|&i| shapes[i] == major).collect();
    for i in 0..clients.len() {
        if shapes[i].is_some() && shapes[i] != major {
            res[i] = verify_rangeproof(clients[i].0, clients[i].1, prove_range).unwrap_or(false);
        }
    }
    let pp: Vec<*const u8> = idx.iter().map(|&i| pbs[i].as_ptr()).collect();
    let cp: Vec<*const u8> = idx.iter().map(|&i| cbs[i].as_ptr()).collect();
    let seed = fresh_seed();
    let mut ok = vec![0 as std::os::raw::c_int; idx.len()];
    let rc = unsafe {
        rofl_verify_rangeproof_batch(idx.len(), pp.as_ptr(), plen, np, cp.as_ptr(), d, prove_range, fp_bits(), fp_frac(), seed.as_ptr(), ok.as_mut_ptr())
    };
    match rc {
        ROFL_OK => { for (k, &i) in idx.iter().enumerate() { res[i] = ok[k] != 0; } Ok(res) }
        ROFL_FORMAT_ERROR => Err(ProofError::FormatError),
        ROFL_INVALID_BITSIZE => Err(ProofError::InvalidBitsize),
        _ => panic!("rofl_zk: {}", last_error()),
    }
}

/// Not in the reference: the updates of several clients hosted by one process (client.rs:265-266 runs them as tasks of one
/// runtime) proved as ONE launch sequence.  Every client's result is bit-identical to `create_rangeproof` for it; a client whose own
/// input is rejected gets its own `Err`, the others are proved.
pub fn create_rangeproof_batch(
    clients: &[(&Vec<f32>, &Vec<Scalar>)],
    prove_range: usize,
    n_partition: usize,
) -> Vec<Result<(Vec<RangeProof>, Vec<RistrettoPoint>), RangeProofError>> {
    if clients.is_empty() { return vec![]; }
    let d = clients[0].0.len();
    assert!(clients.iter().all(|(v, b)| v.len() == d && b.len() == d), "the clients of a batch have d values and d blindings each");
    let bls: Vec<Vec<u8>> = clients.iter().map(|(_, b)| scalars_to_bytes(b)).collect();
    let (n_proofs, plen) = unsafe { (rofl_rangeproof_chunks(d, n_partition), rofl_rangeproof_size(prove_range, d, n_partition)) };
    let mut proofs: Vec<Vec<u8>> = clients.iter().map(|_| vec![0u8; n_proofs * plen]).collect();
    let mut commits: Vec<Vec<u8>> = clients.iter().map(|_| vec![0u8; d * 32]).collect();
    let nonces: Vec<RoflNonce> = clients.iter().map(|_| fresh_nonce()).collect();
    let vp: Vec<*const f32> = clients.iter().map(|(v, _)| v.as_ptr()).collect();
    let bp: Vec<*const u8> = bls.iter().map(|b| b.as_ptr()).collect();
    let pp: Vec<*mut u8> = proofs.iter_mut().map(|p| p.as_mut_ptr()).collect();
    let cp: Vec<*mut u8> = commits.iter_mut().map(|c| c.as_mut_ptr()).collect();
    let mut rcs = vec![0 as std::os::raw::c_int; clients.len()];
    let (mut plen_out, mut n_out) = (0usize, 0usize);
    let rc = unsafe {
        rofl_create_rangeproof_batch(clients.len(), vp.as_ptr(), d, bp.as_ptr(), prove_range, n_partition, fp_bits(), fp_frac(), nonces.as_ptr(),
                                     pp.as_ptr(), &mut plen_out, &mut n_out, cp.as_ptr(), rcs.as_mut_ptr())
    };
    if rc == ROFL_INVALID_BITSIZE { return clients.iter().map(|_| Err(ProofError::InvalidBitsize.into())).collect(); }
    if rc != ROFL_OK { panic!("Should not get here: {}", last_error()); }
    (0..clients.len()).map(|i| match rcs[i] {
        ROFL_OK => Ok((proofs[i][..n_out * plen_out].chunks(plen_out).map(|p| RangeProof::from_bytes(p).expect("librofl_zk proof layout")).collect(),
                       bytes_to_points(&commits[i]))),
        ROFL_VALUE_OUT_OF_RANGE => Err(RangeProofError::ValueOutOfRangeError),
        _ => panic!("Should not get here: client {} of the batch, code {}", i, rcs[i]),
    }).collect()
}
