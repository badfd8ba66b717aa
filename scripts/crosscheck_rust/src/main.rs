//! Pins tests/golden/proofs.json -- proofs made by this repository's oracle and reproduced bit for bit by its HIP kernels -- to the
//! real crates: bulletproofs 4.0.0, curve25519-dalek-ng 4.1.1, merlin 3.0.0, fixed 0.3.3 (and, with `--features sigma`, the
//! reference's own RandProof / SquareRandProof).
//!
//!   cargo run --release [--features sigma] -- ../../tests/golden/proofs.json
//!
//! prints PASS / FAIL per fixture and exits non-zero on any FAIL:
//!   * every range-proof fixture goes through `RangeProof::from_bytes` + `verify_multiple`, composed exactly like
//!     rofl_crypto::range_proof_vec::verify_rangeproof (mod.rs:149-216: shift by 2^(n-1) B, pad with the identity, chunk, label
//!     "RangeProof") resp. l2_range_proof_vec::verify_rangeproof_l2 (mod.rs:185-253: BulletproofGens::new(64, 1), label "L2RangeProof");
//!   * every explicit-stream fixture ("nonce": "stream") is REPLAYED: `prove_multiple_with_rng` with an RNG that hands out the
//!     fixture's bytes must return the same proof bytes and commitments (range_proof_vec/mod.rs:118-142, l2_range_proof_vec/mod.rs:142-183);
//!   * "tie_cases": `fixed` 0.3.3 `saturating_from_float` on values exactly between two grid points (conversion32.rs:11-18) -- the
//!     fixtures assume round-half-to-even;
//!   * `--features sigma`: RandProof / SquareRandProof fixtures through the reference's `verify` (rand_proof/mod.rs:69-91,
//!     square_rand_proof/mod.rs:77-109).  (Their provers draw from thread_rng internally and cannot be replayed.)
use bulletproofs::{BulletproofGens, PedersenGens, RangeProof};
use curve25519_dalek_ng::ristretto::{CompressedRistretto, RistrettoPoint};
use curve25519_dalek_ng::scalar::Scalar;
use curve25519_dalek_ng::traits::Identity;
use merlin::Transcript;
use rand_core::{CryptoRng, Error, RngCore};
use serde_json::Value;

/// Scalar::random(rng) = rng.fill_bytes(&mut [u8; 64]) + from_bytes_mod_order_wide: hand out the fixture's stream, 64 bytes a draw.
struct StreamRng { data: Vec<u8>, pos: usize }
impl RngCore for StreamRng {
    fn next_u32(&mut self) -> u32 { let mut b = [0u8; 4]; self.fill_bytes(&mut b); u32::from_le_bytes(b) }
    fn next_u64(&mut self) -> u64 { let mut b = [0u8; 8]; self.fill_bytes(&mut b); u64::from_le_bytes(b) }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        assert!(self.pos + dest.len() <= self.data.len(), "nonce stream exhausted: the crate draws more than the fixture documents");
        dest.copy_from_slice(&self.data[self.pos..self.pos + dest.len()]);
        self.pos += dest.len();
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), Error> { self.fill_bytes(dest); Ok(()) }
}
impl CryptoRng for StreamRng {}

fn hexv(v: &Value, key: &str) -> Vec<u8> { hex::decode(v[key].as_str().unwrap_or_else(|| panic!("missing {}", key))).unwrap() }
fn usz(v: &Value, key: &str) -> usize { v[key].as_u64().unwrap() as usize }
fn scalar32(b: &[u8]) -> Scalar { let mut a = [0u8; 32]; a.copy_from_slice(b); Scalar::from_canonical_bytes(a).expect("canonical scalar") }
fn next_pow2(val: usize) -> usize { if val <= 1 { 1 } else { val.next_power_of_two() } }

/// conversion32::f32_to_scalar for a run-time (fp_bits, frac): |v| -> Fix::saturating_from_float(..).to_bits(), sign on the scalar.
fn fix_bits(v: f32, fp_bits: usize, frac: usize) -> u64 {
    use fixed::frac::{U12, U3, U7};
    use fixed::{FixedU16, FixedU32, FixedU64, FixedU8};
    let a = v.abs();
    match (fp_bits, frac) {
        (8, 3) => FixedU8::<U3>::saturating_from_float(a).to_bits() as u64,
        (8, 7) => FixedU8::<U7>::saturating_from_float(a).to_bits() as u64,
        (16, 7) => FixedU16::<U7>::saturating_from_float(a).to_bits() as u64,
        (32, 7) => FixedU32::<U7>::saturating_from_float(a).to_bits() as u64,
        (32, 12) => FixedU32::<U12>::saturating_from_float(a).to_bits() as u64,
        (64, 7) => FixedU64::<U7>::saturating_from_float(a).to_bits(),
        _ => panic!("add ({}, {}) to fix_bits", fp_bits, frac),
    }
}
fn f32_to_scalar(v: f32, fp_bits: usize, frac: usize) -> Scalar {
    let s = Scalar::from(fix_bits(v, fp_bits, frac));
    if v < 0.0 { -s } else { s }
}
/// fp::read_from_bytes: the low fp_bits bits of the (shifted) scalar, little endian
fn read_from_bytes(s: &Scalar, fp_bits: usize) -> u64 {
    let mut b = [0u8; 8]; b.copy_from_slice(&s.as_bytes()[..8]);
    let x = u64::from_le_bytes(b);
    if fp_bits >= 64 { x } else { x & ((1u64 << fp_bits) - 1) }
}

/// rofl_crypto::range_proof_vec::verify_rangeproof (mod.rs:149-216) on serialized inputs
fn verify_linf(proofs: &[u8], n_proofs: usize, commits: &[u8], n: usize) -> Result<bool, String> {
    let plen = proofs.len() / n_proofs;
    let pc = PedersenGens::default();
    let off = Scalar::from(1u64 << (n - 1)) * pc.B;
    let mut pts: Vec<RistrettoPoint> = commits.chunks(32).map(|c| CompressedRistretto::from_slice(c).decompress().expect("commitment decompresses") + off).collect();
    let dp = next_pow2(pts.len());
    pts.resize(dp, RistrettoPoint::identity());
    let comp: Vec<CompressedRistretto> = pts.iter().map(|p| p.compress()).collect();
    let m = dp / n_proofs;
    let mut ok = true;
    for (c, pb) in proofs.chunks(plen).enumerate() {
        let proof = RangeProof::from_bytes(pb).map_err(|e| format!("from_bytes: {:?}", e))?;
        let bp = BulletproofGens::new(n, m);
        let mut t = Transcript::new(b"RangeProof");
        ok &= proof.verify_multiple(&bp, &pc, &mut t, &comp[c * m..(c + 1) * m], n).is_ok();
    }
    Ok(ok)
}

/// rofl_crypto::range_proof_vec::create_rangeproof (mod.rs:16-142) with the chunks' nonces replayed from `stream`
fn replay_linf(g: &Value) -> (Vec<u8>, Vec<u8>) {
    let (n, p, fb, ff) = (usz(g, "prove_range"), usz(g, "n_partition"), usz(g, "fp_bits"), usz(g, "fp_frac"));
    let vals: Vec<f32> = g["values"].as_array().unwrap().iter().map(|x| x.as_f64().unwrap() as f32).collect();
    let bl: Vec<Scalar> = hexv(g, "blindings").chunks(32).map(scalar32).collect();
    let stream = hexv(g, "stream");
    let d = vals.len(); let dp = next_pow2(d);
    let off = Scalar::from(1u64 << (n - 1));
    let mut shifted: Vec<u64> = vals.iter().map(|v| read_from_bytes(&(f32_to_scalar(*v, fb, ff) + off), fb)).collect();
    shifted.resize(dp, 0);
    let mut blp = bl.clone(); blp.resize(dp, Scalar::zero());
    let m = dp / std::cmp::min(dp, p);
    let per = m * (2 * n + 4) * 64;
    let pc = PedersenGens::default();
    let (mut proofs, mut commits) = (Vec::new(), Vec::new());
    for c in 0..dp / m {
        let mut rng = StreamRng { data: stream[c * per..(c + 1) * per].to_vec(), pos: 0 };
        let bp = BulletproofGens::new(n, m);
        let mut t = Transcript::new(b"RangeProof");
        let (proof, v) = RangeProof::prove_multiple_with_rng(&bp, &pc, &mut t, &shifted[c * m..(c + 1) * m], &blp[c * m..(c + 1) * m], n, &mut rng).expect("prove");
        assert_eq!(rng.pos, per, "the crate drew {} bytes, the documented order has {}", rng.pos, per);
        proofs.extend_from_slice(&proof.to_bytes());
        for vc in v { commits.push(vc.decompress().unwrap() - off * pc.B); }
    }
    let mut cb = Vec::new();
    for cpt in commits.iter().take(d) { cb.extend_from_slice(cpt.compress().as_bytes()); }
    (proofs, cb)
}

/// rofl_crypto::l2_range_proof_vec::create_rangeproof_l2 (mod.rs:15-183): sum of squares, blinding sum, gens (64, 1), "L2RangeProof"
fn replay_l2(g: &Value) -> (Vec<u8>, Vec<u8>) {
    let (n, fb, ff) = (usz(g, "prove_range"), usz(g, "fp_bits"), usz(g, "fp_frac"));
    let vals: Vec<f32> = g["values"].as_array().unwrap().iter().map(|x| x.as_f64().unwrap() as f32).collect();
    let bl: Vec<Scalar> = hexv(g, "blindings").chunks(32).map(scalar32).collect();
    let val: Scalar = vals.iter().map(|v| { let s = f32_to_scalar(*v, fb, ff); s * s }).sum();
    let bsum: Scalar = bl.iter().sum();
    let mut rng = StreamRng { data: hexv(g, "stream"), pos: 0 };
    let (bp, pc) = (BulletproofGens::new(64, 1), PedersenGens::default());
    let mut t = Transcript::new(b"L2RangeProof");
    let (proof, v) = RangeProof::prove_multiple_with_rng(&bp, &pc, &mut t, &[read_from_bytes(&val, fb)], &[bsum], n, &mut rng).expect("prove");
    (proof.to_bytes(), v[0].as_bytes().to_vec())
}
fn verify_l2(proof: &[u8], commit: &[u8], n: usize) -> Result<bool, String> {
    let p = RangeProof::from_bytes(proof).map_err(|e| format!("from_bytes: {:?}", e))?;
    let (bp, pc) = (BulletproofGens::new(64, 1), PedersenGens::default());
    let mut t = Transcript::new(b"L2RangeProof");
    Ok(p.verify_multiple(&bp, &pc, &mut t, &[CompressedRistretto::from_slice(commit)], n).is_ok())
}

#[cfg(feature = "sigma")]
fn verify_sigma(g: &Value) -> Result<bool, String> {
    use rofl_crypto::rand_proof::{ElGamalGens, ElGamalPair, RandProof};
    use rofl_crypto::square_rand_proof::{pedersen::SquareRandProofCommitments, SquareRandProof};
    let eg = ElGamalGens::default();
    let (pr, cm) = (hexv(g, "proofs"), hexv(g, "commits"));
    let mut ok = true;
    if g["kind"] == "rand" {
        for (p, c) in pr.chunks(128).zip(cm.chunks(64)) {
            let proof = RandProof::from_bytes(p).map_err(|e| format!("{:?}", e))?;
            let pair = ElGamalPair::from_bytes(c).map_err(|e| format!("{:?}", e))?;
            ok &= proof.verify(&eg, &mut Transcript::new(b"RandProof"), pair).is_ok();
        }
    } else {
        for (p, c) in pr.chunks(192).zip(cm.chunks(96)) {
            let proof = SquareRandProof::from_bytes(p).map_err(|e| format!("{:?}", e))?;
            let com = SquareRandProofCommitments::from_bytes(c).map_err(|e| format!("{:?}", e))?;
            ok &= proof.verify(&eg, &mut Transcript::new(b"SquareRandProof"), com).is_ok();
        }
    }
    Ok(ok)
}
#[cfg(not(feature = "sigma"))]
fn verify_sigma(_g: &Value) -> Result<bool, String> { Err("skipped (build with --features sigma and ./rofl_crypto)".into()) }

fn main() {
    let path = std::env::args().nth(1).unwrap_or_else(|| "../../tests/golden/proofs.json".into());
    let fixtures: Value = serde_json::from_str(&std::fs::read_to_string(&path).expect("read fixtures")).expect("json");
    let (mut pass, mut fail, mut skip) = (0, 0, 0);
    let mut report = |name: String, r: Result<bool, String>| match r {
        Ok(true) => { println!("PASS  {}", name); pass += 1; }
        Ok(false) => { println!("FAIL  {}", name); fail += 1; }
        Err(e) if e.starts_with("skipped") => { println!("SKIP  {}: {}", name, e); skip += 1; }
        Err(e) => { println!("FAIL  {}: {}", name, e); fail += 1; }
    };
    for (i, g) in fixtures.as_array().unwrap().iter().enumerate() {
        let kind = g["kind"].as_str().unwrap();
        let stream = g["nonce"] == "stream";
        let tag = format!("#{} {}{}", i, kind, if stream { " (explicit nonce stream)" } else { "" });
        match kind {
            "linf" => {
                let (n, np) = (usz(g, "prove_range"), usz(g, "n_proofs"));
                report(format!("{} n={} d={} verify_multiple", tag, n, usz(g, "d")), verify_linf(&hexv(g, "proofs"), np, &hexv(g, "commits"), n));
                let mut bad = hexv(g, "proofs"); bad[5 * 32 + 3] ^= 1;
                report(format!("{} tampered proof is rejected", tag), verify_linf(&bad, np, &hexv(g, "commits"), n).map(|ok| !ok));
                if stream {
                    let (p, c) = replay_linf(g);
                    report(format!("{} prove_multiple_with_rng reproduces the proof bytes", tag), Ok(p == hexv(g, "proofs")));
                    report(format!("{} ... and the commitments", tag), Ok(c == hexv(g, "commits")));
                }
            }
            "l2" => {
                report(format!("{} verify_multiple", tag), verify_l2(&hexv(g, "proofs"), &hexv(g, "commits"), usz(g, "prove_range")));
                if stream {
                    let (p, c) = replay_l2(g);
                    report(format!("{} prove_multiple_with_rng reproduces proof and commitment", tag), Ok(p == hexv(g, "proofs") && c == hexv(g, "commits")));
                }
            }
            "rand" | "sqrand" => report(format!("{} reference verify", tag), verify_sigma(g)),
            "tie_cases" => {
                for c in g["cases"].as_array().unwrap() {
                    let v = c["v"].as_f64().unwrap() as f32;
                    let (fb, ff) = (usz(c, "fp_bits"), usz(c, "fp_frac"));
                    let got = fix_bits(v, fb, ff);
                    let want = c["bits_half_even"].as_u64().unwrap();
                    let note = if got == want { String::new() } else { format!(" -- fixed 0.3.3 gives {} (half away from zero would be {}): the oracle's tie rule must change", got, c["bits_half_away"]) };
                    report(format!("tie v={} fp{}/frac{}{}", v, fb, ff, note), Ok(got == want && f32_to_scalar(v, fb, ff).as_bytes()[..] == hex::decode(c["scalar"].as_str().unwrap()).unwrap()[..]));
                }
            }
            other => println!("SKIP  #{} unknown kind {}", i, other),
        }
    }
    println!("{} passed, {} failed, {} skipped", pass, fail, skip);
    std::process::exit(if fail == 0 { 0 } else { 1 });
}
