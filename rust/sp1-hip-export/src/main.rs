//! `sp1-hip-export`: every chip of an SP1 machine as DATA — the interchange document `libsp1hip.so` proves from
//! (schema: sp1_amd/machine.py of the backend repository; consumer: `sp1_hip_prover::MachineDescription`).
//!
//! UNCOMPILED in the repository that produced it. The same trick the reference uses for its CUDA backend: run every chip's
//! `Air::eval` once over a builder that records instead of computing (sp1-gpu/crates/air/src/ir/builder.rs:L29-L66 is the
//! reference's `DagBuilder`; crates/core/compiler/src/main.rs:L11-L60 its command-line front end), and walk
//! `chip.sends()` / `chip.receives()` for the lookups.
//!
//!   sp1-hip-export --machine core     > rv64im_core.json        (RiscvAir::machine())
//!   sp1-hip-export --machine compress > recursion_compress.json  (CompressAir::compress_machine())
//!   sp1-hip-export --machine shrink   > recursion_shrink.json    (CompressAir::shrink_machine())
//!
//! Acceptance test: the `compress` document must equal sp1_amd/machines/recursion_compress.json as a POLYNOMIAL system —
//! that file is a hand transcription of the same machine which the reference's own compress proof verifies against
//!; instruction numbering may differ, so the check is tests/test_rust_glue.py's evaluator, not a diff.
mod recorder;

use clap::{Parser, ValueEnum};
use serde_json::{json, Value};
use slop_air::{Air, PairCol, VirtualPairCol};
use slop_algebra::PrimeField32;
use sp1_core_machine::riscv::RiscvAir;
use sp1_hypercube::{air::MachineAir, Interaction, Machine};
use sp1_primitives::SP1Field;
use sp1_prover::CompressAir;

use recorder::{record, RecordingBuilder};

type F = SP1Field;

#[derive(ValueEnum, Clone, Debug)]
enum Which {
    Core,
    Compress,
    Shrink,
}

#[derive(Parser, Debug)]
#[command(about = "Export an SP1 machine's chips for libsp1hip.so")]
struct Args {
    #[arg(long, value_enum)]
    machine: Which,
}

fn vcol(c: &VirtualPairCol<F>) -> Value {
    // `VirtualPairCol { column_weights: Vec<(PairCol, F)>, constant: F }` = sum weight * column + constant
    let terms: Vec<Value> = c
        .column_weights()
        .iter()
        .map(|(col, w)| match col {
            PairCol::Main(i) => json!(["main", i, w.as_canonical_u32()]),
            PairCol::Preprocessed(i) => json!(["prep", i, w.as_canonical_u32()]),
        })
        .collect();
    json!({ "constant": c.constant().as_canonical_u32(), "terms": terms })
}

fn interaction(i: &Interaction<F>) -> Value {
    json!({
        "kind": i.argument_index(),
        "multiplicity": vcol(&i.multiplicity),
        "values": i.values.iter().map(vcol).collect::<Vec<_>>(),
    })
}

fn dump<A>(machine: &Machine<F, A>) -> Value
where
    A: MachineAir<F> + for<'a> Air<RecordingBuilder<'a>>,
{
    let mut chips: Vec<Value> = machine
        .chips()
        .iter()
        .map(|chip| {
            // the k-th `assert_zero` is constraint k: the folder multiplies it by alpha^(K - 1 - k)
            // (crates/hypercube/src/folder.rs:L276-L323)
            let instrs = record(chip.air.as_ref(), chip.preprocessed_width(), chip.width(), machine.num_pv_elts());
            json!({
                "name": chip.name(),
                "main_width": chip.width(),
                "preprocessed_width": chip.preprocessed_width(),
                "constraints": instrs,
                "sends": chip.sends().iter().map(interaction).collect::<Vec<_>>(),
                "receives": chip.receives().iter().map(interaction).collect::<Vec<_>>(),
            })
        })
        .collect();
    // `BTreeSet<Chip>` order = name order: what the transcript and the commitment use
    chips.sort_by(|a, b| a["name"].as_str().cmp(&b["name"].as_str()));
    json!({ "field": "KoalaBear", "chips": chips })
}

fn main() {
    let args = Args::parse();
    let doc = match args.machine {
        Which::Core => dump(&RiscvAir::<F>::machine()),
        Which::Compress => dump(&CompressAir::<F>::compress_machine()),
        Which::Shrink => dump(&CompressAir::<F>::shrink_machine()),
    };
    println!("{}", serde_json::to_string(&doc).unwrap());
}
