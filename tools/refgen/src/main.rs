//! refgen: golden vectors from the reference implementation itself.
//!
//! Writes, into the directory given as first argument (default `.`):
//!
//!   ref_g1_msm.json       `NativeLoader::multi_scalar_multiplication` (snark-verifier/src/loader/native.rs:61-71) for
//!                         n in {1, 2, 3, 21, 64, 65, 1024} and `util::msm::multi_scalar_multiplication`
//!                         (util/msm.rs:308-343, `.to_affine()`) at 2^10 and 2^16 -- schema of tests/golden/g1_msm.json
//!   ref_kzg_as.json       `KzgAs::create_proof` (non-zk, fresh EvmTranscript; pcs/kzg/accumulation.rs:148-197) over 64
//!                         valid accumulators -- {accumulators, result}
//!   ref_kzg_as_poseidon.json  the same call over a fresh PoseidonTranscript (T = 5, RATE = 4, R_F = 8, R_P = 60), the
//!                         transcript the reference's example uses for its accumulation proof
//!                         (examples/evm-verifier-with-accumulator.rs:375) -- {accumulators, result}
//!   ref_kzg_decider.json  `KzgAs::decide` (pcs/kzg/decider.rs:70-82) accept / reject cases -- schema of kzg_decider.json
//!   ref_limbs.json        `LimbsEncoding::<4, 68>` / `fe_to_limbs` (pcs/kzg/accumulator.rs:57-81, util/arithmetic.rs:286-298)
//!   ref_snark.bin / .json a REAL halo2 proof: bincode / serde_json of the SDK's `Snark { protocol, instances, proof }`
//!                         (snark-verifier-sdk/src/lib.rs:47-53; the struct is restated here with the same fields in the
//!                         same order, which is all serde's derive looks at)
//!   ref_snark_meta.json   deciding key bytes, the accumulator `PlonkSuccinctVerifier::verify` returns for that proof,
//!                         the verdict of `PlonkVerifier::verify`
//!
//! Byte conventions = this repository's C ABI: Fr / Fq 32 B little-endian canonical (`to_repr`), G1 = x | y,
//! G2 = x.c0 | x.c1 | y.c0 | y.c1, accumulator = lhs | rhs, everything hex in the JSON files.
//! Inputs are drawn from ChaCha20 with fixed seeds, so every run writes the same files.
use halo2_proofs::{
    circuit::{Layouter, SimpleFloorPlanner, Value},
    plonk::{create_proof, keygen_pk, keygen_vk, Advice, Circuit, Column, ConstraintSystem, Error, Fixed, Instance},
    poly::{
        commitment::{Params, ParamsProver},
        kzg::{
            commitment::{KZGCommitmentScheme, ParamsKZG},
            multiopen::ProverGWC,
        },
        Rotation,
    },
    transcript::TranscriptWriterBuffer,
};
use halo2curves::{
    bn256::{Bn256, Fq, Fr, G1Affine, G2Affine, G1},
    ff::{Field, PrimeField},
    group::{prime::PrimeCurveAffine, Curve, Group},
    CurveAffine,
};
use rand::SeedableRng;
use rand_chacha::ChaCha20Rng;
use serde_json::json;
use snark_verifier::{
    loader::{native::NativeLoader, EcPointLoader},
    pcs::{
        kzg::{Gwc19, KzgAccumulator, KzgAs, KzgAsProvingKey, KzgDecidingKey, LimbsEncoding},
        AccumulationDecider, AccumulationSchemeProver, AccumulatorEncoding,
    },
    system::halo2::{
        compile,
        transcript::{evm::EvmTranscript, halo2::PoseidonTranscript},
        Config,
    },
    util::{arithmetic::fe_to_limbs, msm::multi_scalar_multiplication},
    verifier::{
        plonk::{PlonkProtocol, PlonkSuccinctVerifier, PlonkVerifier},
        SnarkVerifier,
    },
};
use std::{fs, path::PathBuf};

type As = KzgAs<Bn256, Gwc19>;

fn fr_hex(x: &Fr) -> String {
    hex::encode(x.to_repr())
}
fn g1_bytes(p: &G1Affine) -> Vec<u8> {
    // identity = 64 zero bytes (this repository's convention; `coordinates()` is None for it)
    let mut out = vec![0u8; 64];
    if let Some(c) = Option::<halo2curves::Coordinates<G1Affine>>::from(p.coordinates()) {
        out[..32].copy_from_slice(c.x().to_repr().as_ref());
        out[32..].copy_from_slice(c.y().to_repr().as_ref());
    }
    out
}
fn g2_bytes(p: &G2Affine) -> Vec<u8> {
    let c = Option::<halo2curves::Coordinates<G2Affine>>::from(p.coordinates()).expect("G2 point at infinity");
    let (x, y) = (c.x(), c.y());
    let mut out = Vec::with_capacity(128);
    for f in [&x.c0, &x.c1, &y.c0, &y.c1] {
        out.extend_from_slice(Fq::to_repr(f).as_ref());
    }
    out
}
fn acc_bytes(a: &KzgAccumulator<G1Affine, NativeLoader>) -> Vec<u8> {
    [g1_bytes(&a.lhs), g1_bytes(&a.rhs)].concat()
}

/// `KzgAs::decide` on the native loader (pcs/kzg/decider.rs:70-82); fully qualified: the EVM loader has an impl too
fn decide(dk: &KzgDecidingKey<Bn256>, acc: KzgAccumulator<G1Affine, NativeLoader>) -> bool {
    <As as AccumulationDecider<G1Affine, NativeLoader>>::decide(dk, acc).is_ok()
}

fn msm_cases(rng: &mut ChaCha20Rng) -> serde_json::Value {
    let mut cases = Vec::new();
    for n in [1usize, 2, 3, 21, 64, 65, 1024] {
        let scalars: Vec<Fr> = (0..n).map(|_| Fr::random(&mut *rng)).collect();
        let points: Vec<G1Affine> = (0..n).map(|_| G1::random(&mut *rng).to_affine()).collect();
        let pairs: Vec<(&Fr, &G1Affine)> = scalars.iter().zip(points.iter()).collect();
        let out = <NativeLoader as EcPointLoader<G1Affine>>::multi_scalar_multiplication(&pairs);
        cases.push(json!({
            "name": format!("native_loader_n{n}"), "api": "loader::native::NativeLoader::multi_scalar_multiplication",
            "scalars": scalars.iter().map(fr_hex).collect::<String>(),
            "points": hex::encode(points.iter().flat_map(g1_bytes).collect::<Vec<u8>>()),
            "expected": hex::encode(g1_bytes(&out)),
        }));
    }
    for log2n in [10u32, 16] {
        let n = 1usize << log2n;
        let scalars: Vec<Fr> = (0..n).map(|_| Fr::random(&mut *rng)).collect();
        let points: Vec<G1Affine> = (0..n).map(|_| G1::random(&mut *rng).to_affine()).collect();
        let out = multi_scalar_multiplication(&scalars, &points).to_affine();
        cases.push(json!({
            "name": format!("util_msm_2p{log2n}"), "api": "util::msm::multi_scalar_multiplication",
            "scalars": scalars.iter().map(fr_hex).collect::<String>(),
            "points": hex::encode(points.iter().flat_map(g1_bytes).collect::<Vec<u8>>()),
            "expected": hex::encode(g1_bytes(&out)),
        }));
    }
    json!({"generator": "tools/refgen", "oracle": "snark-verifier + halo2curves 0.6.0 (the reference itself)", "cases": cases})
}

/// A small original circuit (not the reference's example): one advice column, one fixed selector, one instance column;
/// gate q * (a * a - a_next) = 0 on row 0, a_next exposed as the public input.
#[derive(Clone, Default)]
struct Square(Fr);

#[derive(Clone)]
struct SquareConfig {
    a: Column<Advice>,
    q: Column<Fixed>,
    inst: Column<Instance>,
}

impl Circuit<Fr> for Square {
    type Config = SquareConfig;
    type FloorPlanner = SimpleFloorPlanner;
    fn without_witnesses(&self) -> Self {
        Self::default()
    }
    fn configure(meta: &mut ConstraintSystem<Fr>) -> SquareConfig {
        let (a, q, inst) = (meta.advice_column(), meta.fixed_column(), meta.instance_column());
        meta.enable_equality(a);
        meta.enable_equality(inst);
        meta.create_gate("q * (a^2 - a_next) = 0", |m| {
            let (a0, a1, q) = (m.query_advice(a, Rotation::cur()), m.query_advice(a, Rotation::next()), m.query_fixed(q, Rotation::cur()));
            vec![q * (a0.clone() * a0 - a1)]
        });
        SquareConfig { a, q, inst }
    }
    fn synthesize(&self, cfg: SquareConfig, mut layouter: impl Layouter<Fr>) -> Result<(), Error> {
        let out = layouter.assign_region(
            || "square",
            |mut region| {
                region.assign_fixed(|| "q", cfg.q, 0, || Value::known(Fr::ONE))?;
                region.assign_advice(|| "a", cfg.a, 0, || Value::known(self.0))?;
                region.assign_advice(|| "a^2", cfg.a, 1, || Value::known(self.0.square()))
            },
        )?;
        layouter.constrain_instance(out.cell(), cfg.inst, 0)
    }
}

/// field order and names of snark-verifier-sdk/src/lib.rs:47-53
#[derive(serde::Serialize)]
struct Snark {
    protocol: PlonkProtocol<G1Affine>,
    instances: Vec<Vec<Fr>>,
    proof: Vec<u8>,
}

fn main() {
    let dir = PathBuf::from(std::env::args().nth(1).unwrap_or_else(|| ".".into()));
    fs::create_dir_all(&dir).unwrap();
    let write = |name: &str, v: &serde_json::Value| fs::write(dir.join(name), serde_json::to_vec_pretty(v).unwrap()).unwrap();
    let mut rng = ChaCha20Rng::seed_from_u64(0x5EED_0001);

    write("ref_g1_msm.json", &msm_cases(&mut rng));

    // ---- KZG layer under a toy SRS secret
    let s = Fr::random(&mut rng);
    let (g1, g2) = (G1Affine::generator(), G2Affine::generator());
    let s_g2 = (g2 * s).to_affine();
    let dk = KzgDecidingKey::<Bn256>::new(g1, g2, s_g2);
    let valid = |rng: &mut ChaCha20Rng| {
        let rhs = G1::random(&mut *rng).to_affine();
        KzgAccumulator::<G1Affine, NativeLoader>::new((rhs * s).to_affine(), rhs)
    };
    let accs: Vec<_> = (0..64).map(|_| valid(&mut rng)).collect();
    let mut t = EvmTranscript::<G1Affine, NativeLoader, _, _>::new(Vec::new());
    let folded = As::create_proof(&KzgAsProvingKey::new(None), &accs, &mut t, &mut rng).unwrap();
    write("ref_kzg_as.json", &json!({
        "generator": "tools/refgen", "api": "KzgAs::<Bn256, Gwc19>::create_proof (non-zk, EvmTranscript over an empty stream)",
        "accumulators": hex::encode(accs.iter().flat_map(acc_bytes).collect::<Vec<u8>>()),
        "result": hex::encode(acc_bytes(&folded)),
    }));

    // ... and over a fresh POSEIDON transcript, as the reference's own example writes its accumulation proof
    // (examples/evm-verifier-with-accumulator.rs:36-39,375: T = 5, RATE = 4, R_F = 8, R_P = 60).  This is the vector that pins
    // what cannot be read off the reference's sources: the external `poseidon` crate's round constants / MDS and
    // `State::default()`, and how a G1 point's coordinates enter the sponge (`fe_to_fe`, halo2.rs:220-236).
    let mut tp = PoseidonTranscript::<G1Affine, NativeLoader, _, 5, 4, 8, 60>::new(Vec::new());
    let folded_poseidon = As::create_proof(&KzgAsProvingKey::new(None), &accs, &mut tp, &mut rng).unwrap();
    write("ref_kzg_as_poseidon.json", &json!({
        "generator": "tools/refgen",
        "api": "KzgAs::<Bn256, Gwc19>::create_proof (non-zk, PoseidonTranscript<_, NativeLoader, _, 5, 4, 8, 60> over an empty stream)",
        "accumulators": hex::encode(accs.iter().flat_map(acc_bytes).collect::<Vec<u8>>()),
        "result": hex::encode(acc_bytes(&folded_poseidon)),
    }));

    let mut cases = Vec::new();
    for i in 0..4 {
        let a = valid(&mut rng);
        cases.push(json!({"name": format!("valid_{i}"), "acc": hex::encode(acc_bytes(&a)), "accept": decide(&dk, a)}));
    }
    cases.push(json!({"name": "folded_64", "acc": hex::encode(acc_bytes(&folded)), "accept": decide(&dk, folded.clone())}));
    for i in 0..4 {
        let mut a = valid(&mut rng);
        a.lhs = (a.lhs.to_curve() + G1::generator()).to_affine();
        cases.push(json!({"name": format!("invalid_{i}"), "acc": hex::encode(acc_bytes(&a)), "accept": decide(&dk, a)}));
    }
    write("ref_kzg_decider.json", &json!({
        "generator": "tools/refgen", "oracle": "halo2curves 0.6.0 bn256 pairing via KzgAs::decide",
        "g1": hex::encode(g1_bytes(&g1)), "g2": hex::encode(g2_bytes(&g2)), "s_g2": hex::encode(g2_bytes(&s_g2)),
        "secret": fr_hex(&s), "cases": cases,
    }));

    let limbs: Vec<Fr> = [folded.lhs, folded.rhs]
        .iter()
        .flat_map(|p| {
            let c = p.coordinates().unwrap();
            [*c.x(), *c.y()]
        })
        .flat_map(|fe: Fq| fe_to_limbs::<Fq, Fr, 4, 68>(fe))
        .collect();
    let back = <LimbsEncoding<4, 68> as AccumulatorEncoding<G1Affine, NativeLoader>>::from_repr(&limbs.iter().collect::<Vec<_>>()).unwrap();
    assert_eq!(acc_bytes(&back), acc_bytes(&folded));
    write("ref_limbs.json", &json!({
        "generator": "tools/refgen", "accumulator": hex::encode(acc_bytes(&folded)),
        "limbs": limbs.iter().map(fr_hex).collect::<String>(),
    }));

    // ---- a real proof: keygen, prove (GWC19, EvmTranscript), compile the protocol, verify with the reference
    let k = 6u32;
    let params = ParamsKZG::<Bn256>::setup(k, &mut rng);
    let circuit = Square(Fr::random(&mut rng));
    let instances = vec![vec![circuit.0.square()]];
    let pk = keygen_pk(&params, keygen_vk(&params, &circuit).unwrap(), &circuit).unwrap();
    let proof = {
        let inst: Vec<&[Fr]> = instances.iter().map(|v| v.as_slice()).collect();
        let mut tw = TranscriptWriterBuffer::<_, G1Affine, _>::init(Vec::new());
        create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC<_>, _, _, EvmTranscript<_, _, _, _>, _>(
            &params, &pk, &[circuit], &[inst.as_slice()], &mut rng, &mut tw,
        )
        .unwrap();
        tw.finalize()
    };
    let protocol = compile(&params, pk.get_vk(), Config::kzg().with_num_instance(vec![1]));
    let dk: KzgDecidingKey<Bn256> = (params.get_g()[0], params.g2(), params.s_g2()).into();
    let mut tr = EvmTranscript::<G1Affine, NativeLoader, _, _>::new(proof.as_slice());
    let pf = PlonkSuccinctVerifier::<As>::read_proof(&dk.svk, &protocol, &instances, &mut tr).unwrap();
    let accs = PlonkSuccinctVerifier::<As>::verify(&dk.svk, &protocol, &instances, &pf).unwrap();
    let mut tr = EvmTranscript::<G1Affine, NativeLoader, _, _>::new(proof.as_slice());
    let pf2 = PlonkVerifier::<As>::read_proof(&dk, &protocol, &instances, &mut tr).unwrap();
    let accepted = PlonkVerifier::<As>::verify(&dk, &protocol, &instances, &pf2).is_ok();
    let snark = Snark { protocol, instances, proof };
    fs::write(dir.join("ref_snark.bin"), bincode::serialize(&snark).unwrap()).unwrap();
    fs::write(dir.join("ref_snark.json"), serde_json::to_vec(&snark).unwrap()).unwrap();
    write("ref_snark_meta.json", &json!({
        "generator": "tools/refgen", "mos": "gwc19", "transcript": "evm", "k": k,
        "dk": hex::encode([g1_bytes(&dk.svk.g), g2_bytes(&dk.g2), g2_bytes(&dk.s_g2)].concat()),
        "accumulators": hex::encode(accs.iter().flat_map(acc_bytes).collect::<Vec<u8>>()),
        "accepted": accepted,
    }));
    println!("refgen: wrote ref_g1_msm.json ref_kzg_as.json ref_kzg_decider.json ref_limbs.json ref_snark.bin ref_snark.json ref_snark_meta.json into {}", dir.display());
}
