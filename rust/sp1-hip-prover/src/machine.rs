//! The chips of a machine as DATA: the JSON interchange document written by `sp1-hip-export` (schema in
//! sp1_amd/machine.py of the backend repository) turned into the word arrays `sp1hip_shard_chip_t` takes.
//!
//!   constraints : `[op, a, b]` SSA triples (0 LOAD_MAIN col | 1 LOAD_PREP col | 2 CONST canonical | 3 PUBLIC idx |
//!                 4 ADD | 5 SUB | 6 MUL | 7 NEG | 8 ASSERT_ZERO) — the data form of `Air::eval` over
//!                 `ConstraintSumcheckFolder` (/root/reference/crates/hypercube/src/folder.rs:L276-L323)
//!   interactions: `[n, per interaction: is_send, kind, n_values, vcol(multiplicity), vcol(value)...]`,
//!                 `vcol = [n_terms, constant, n_terms x (is_main, column, weight)]` — the data form of
//!                 `Interaction { values, multiplicity, kind }` (/root/reference/crates/hypercube/src/lookup/interaction.rs:L11-L24),
//!                 sends first, then receives (crates/hypercube/src/logup_gkr/cpu.rs:L86-L92)
use std::{collections::BTreeMap, ffi::CString};

use serde::Deserialize;

use crate::error::HipError;

const P: u64 = 0x7f00_0001;

#[derive(Deserialize)]
struct VColDoc {
    #[serde(default)]
    constant: u64,
    #[serde(default)]
    terms: Vec<(String, u32, u64)>,
}

#[derive(Deserialize)]
struct InteractionDoc {
    kind: u32,
    multiplicity: VColDoc,
    values: Vec<VColDoc>,
}

#[derive(Deserialize)]
struct ChipDoc {
    name: String,
    main_width: u32,
    #[serde(default)]
    preprocessed_width: u32,
    constraints: Vec<[u32; 3]>,
    #[serde(default)]
    sends: Vec<InteractionDoc>,
    #[serde(default)]
    receives: Vec<InteractionDoc>,
}

#[derive(Deserialize)]
struct MachineDoc {
    #[serde(default = "koala")]
    field: String,
    chips: Vec<ChipDoc>,
}
fn koala() -> String {
    "KoalaBear".into()
}

/// One chip's programs in the layouts of include/sp1hip.h.
pub struct ChipProgram {
    pub name: CString,
    pub main_width: u32,
    pub prep_width: u32,
    /// `[n_instr][3]` constraint program, flattened.
    pub constraints: Vec<u32>,
    pub num_constraints: u32,
    /// interaction program words.
    pub interactions: Vec<u32>,
}

/// Every chip of a machine, by name (`BTreeMap` = the `BTreeSet<Chip>` order the transcript uses).
pub struct MachineDescription {
    pub chips: BTreeMap<String, ChipProgram>,
}

fn vcol_words(v: &VColDoc, mw: u32, pw: u32, out: &mut Vec<u32>, at: &str) -> Result<(), HipError> {
    if v.constant >= P {
        return Err(HipError::Description(format!("{at}: constant is not a canonical field element")));
    }
    out.push(v.terms.len() as u32);
    out.push(v.constant as u32);
    for (kind, col, weight) in &v.terms {
        let (is_main, width) = match kind.as_str() {
            "main" => (1u32, mw),
            "prep" => (0u32, pw),
            other => return Err(HipError::Description(format!("{at}: column kind {other:?}"))),
        };
        if *col >= width || *weight >= P {
            return Err(HipError::Description(format!("{at}: column {col} / weight {weight} out of range")));
        }
        out.extend_from_slice(&[is_main, *col, *weight as u32]);
    }
    Ok(())
}

impl MachineDescription {
    /// Parse and validate the interchange document (the same checks as `load_machine` of sp1_amd/machine.py).
    pub fn from_json(doc: &str) -> Result<Self, HipError> {
        let doc: MachineDoc = serde_json::from_str(doc).map_err(|e| HipError::Description(e.to_string()))?;
        if doc.field != "KoalaBear" {
            return Err(HipError::Description(format!("field {:?}: only KoalaBear machines are supported", doc.field)));
        }
        let mut chips = BTreeMap::new();
        for c in doc.chips {
            let (mw, pw) = (c.main_width, c.preprocessed_width);
            let mut num_constraints = 0u32;
            for (k, [op, a, b]) in c.constraints.iter().copied().enumerate() {
                let k = k as u32;
                let ok = match op {
                    0 => a < mw,
                    1 => a < pw,
                    2 => (a as u64) < P,
                    3 => true,
                    4..=6 => a < k && b < k,
                    7 => a < k,
                    8 => {
                        num_constraints += 1;
                        a < k
                    }
                    _ => false,
                };
                if !ok {
                    return Err(HipError::Description(format!("{} constraint instruction {k}: bad opcode / operand", c.name)));
                }
            }
            let mut words = vec![(c.sends.len() + c.receives.len()) as u32];
            for (is_send, list) in [(1u32, &c.sends), (0u32, &c.receives)] {
                for (j, it) in list.iter().enumerate() {
                    let at = format!("{} {}[{j}]", c.name, if is_send == 1 { "sends" } else { "receives" });
                    words.extend_from_slice(&[is_send, it.kind, it.values.len() as u32]);
                    vcol_words(&it.multiplicity, mw, pw, &mut words, &at)?;
                    for v in &it.values {
                        vcol_words(v, mw, pw, &mut words, &at)?;
                    }
                }
            }
            let name = CString::new(c.name.clone()).map_err(|e| HipError::Description(e.to_string()))?;
            let prog = ChipProgram {
                name,
                main_width: mw,
                prep_width: pw,
                constraints: c.constraints.iter().flatten().copied().collect(),
                num_constraints,
                interactions: words,
            };
            if chips.insert(c.name.clone(), prog).is_some() {
                return Err(HipError::Description(format!("chip {:?} appears twice", c.name)));
            }
        }
        Ok(Self { chips })
    }
}
