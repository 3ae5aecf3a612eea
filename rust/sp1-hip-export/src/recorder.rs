//! A recording `AirBuilder`: running a chip's `eval` against it yields the chip's constraints as an SSA program
//! `[op, a, b]` (instruction k defines value k) in the opcode set of include/sp1hip.h:
//!
//!   0 LOAD_MAIN col | 1 LOAD_PREP col | 2 CONST canonical | 3 PUBLIC idx | 4 ADD a b | 5 SUB a b | 6 MUL a b | 7 NEG a |
//!   8 ASSERT_ZERO a | 16 HINT kind col (optional pseudo-instruction, no value: kind 1 = "the next 163 asserts are the
//!   Poseidon2 permutation sub-AIR over main columns [col, col + 179)", what `hint_poseidon2` below records when a chip's
//!   eval enters `eval_external_round` / `eval_internal_rounds`; kinds 2 / 3 = the septic curve equation / sum checker of the
//!   Global chip, `hint_septic_curve` / `hint_septic_sum`; the HIP prover evaluates such a block with a fused kernel)
//!
//! Same job as the reference's `DagBuilder` (sp1-gpu/crates/air/src/ir/builder.rs:L29-L66, expr.rs, var.rs), different
//! design: no global DAG behind a mutex, no node enum — a thread-local instruction list with hash-consing (a repeated load /
//! constant / operation reuses the earlier value; ADD and MUL are commutative), which is what keeps Poseidon2's thousands of
//! repeated sub-expressions small. Single-row constraints only: the zerocheck folder the prover implements exposes no
//! next-row access (crates/hypercube/src/folder.rs:L276-L323), so `is_first_row` / `is_last_row` / `is_transition` are
//! rejected exactly as the reference's builder rejects them. UNCOMPILED in the repository that produced it.
use std::{
    cell::RefCell,
    collections::HashMap,
    iter::{Product, Sum},
    ops::{Add, AddAssign, Mul, MulAssign, Neg, Sub, SubAssign},
};

use slop_air::{Air, AirBuilder, AirBuilderWithPublicValues, ExtensionBuilder, PairBuilder, PermutationAirBuilder};
use slop_algebra::{AbstractExtensionField, AbstractField, PrimeField32};
use slop_matrix::dense::{DenseMatrix, RowMajorMatrixView};
use sp1_core_machine::air::TrivialOperationBuilder;
use sp1_hypercube::air::{EmptyMessageBuilder, MachineAir};
use sp1_primitives::{SP1ExtensionField, SP1Field};

type F = SP1Field;
type EF = SP1ExtensionField;

const LOAD_MAIN: u32 = 0;
const LOAD_PREP: u32 = 1;
const CONST: u32 = 2;
const PUBLIC: u32 = 3;
const ADD: u32 = 4;
const SUB: u32 = 5;
const MUL: u32 = 6;
const NEG: u32 = 7;
const ASSERT_ZERO: u32 = 8;
const HINT: u32 = 16;
const HINT_POSEIDON2: u32 = 1;

#[derive(Default)]
struct Tape {
    instrs: Vec<[u32; 3]>,
    seen: HashMap<[u32; 3], u32>,
}

impl Tape {
    fn emit(&mut self, op: u32, a: u32, b: u32) -> u32 {
        if op == ASSERT_ZERO {
            self.instrs.push([op, a, b]);
            return self.instrs.len() as u32 - 1;
        }
        let key = if op == ADD || op == MUL { [op, a.min(b), a.max(b)] } else { [op, a, b] };
        if let Some(&k) = self.seen.get(&key) {
            return k;
        }
        self.instrs.push([op, a, b]);
        let k = self.instrs.len() as u32 - 1;
        self.seen.insert(key, k);
        k
    }
}

thread_local! {
    static TAPE: RefCell<Tape> = RefCell::new(Tape::default());
}

fn emit(op: u32, a: u32, b: u32) -> u32 {
    TAPE.with(|t| t.borrow_mut().emit(op, a, b))
}

/// Marks the start of a Poseidon2 permutation sub-AIR (the operation's columns start at main column `first_col`): the
/// exporter calls this from its `SP1OperationBuilder` hook for the chips that embed a `Poseidon2Operation` (Global,
/// Poseidon2Wide) right before lowering the operation. Never merged, defines no value.
pub fn hint_poseidon2(first_col: u32) {
    TAPE.with(|t| t.borrow_mut().instrs.push([HINT, HINT_POSEIDON2, first_col]));
}

/// The 7 asserts that follow are the septic curve equation y^2 = x^3 + 45 x + 41 z^3 over main columns [xy_col, xy_col + 14)
/// (x then y; `GlobalInteractionOperation::eval_single_digest`, operations/global_interaction.rs:L203-L208). Kind 2.
pub fn hint_septic_curve(xy_col: u32) {
    TAPE.with(|t| t.borrow_mut().instrs.push([HINT, 2, xy_col]));
}

/// The 14 asserts that follow are `sum_checker_x` and `is_real * sum_checker_y` of the global accumulation
/// (operations/global_accumulation.rs:L83-L131) for p1 = main columns [acc_col, +14), p2 = [xy_col, +14), p3 = [acc_col + 14, +14).
/// Kind 3; the operand words carry the `is_real` column (bits 8.. of the first) and both column bases (16 bits each).
pub fn hint_septic_sum(xy_col: u32, acc_col: u32, is_real_col: u32) {
    assert!(xy_col < (1 << 16) && acc_col < (1 << 16) && is_real_col < (1 << 24));
    TAPE.with(|t| t.borrow_mut().instrs.push([HINT, 3 | (is_real_col << 8), xy_col | (acc_col << 16)]));
}

fn constant(f: F) -> u32 {
    emit(CONST, f.as_canonical_u32(), 0)
}

/// A value of the program (the index of the instruction that defines it). `Var` and `Expr` are the same thing here; two
/// types only because `AirBuilder` wants them distinct.
#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
pub struct Var(pub u32);
#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
pub struct Expr(pub u32);

impl From<Var> for Expr {
    fn from(v: Var) -> Self {
        Expr(v.0)
    }
}
impl From<F> for Expr {
    fn from(f: F) -> Self {
        Expr(constant(f))
    }
}
impl Default for Expr {
    fn default() -> Self {
        Self::zero()
    }
}

trait Operand {
    fn id(self) -> u32;
}
impl Operand for Var {
    fn id(self) -> u32 {
        self.0
    }
}
impl Operand for Expr {
    fn id(self) -> u32 {
        self.0
    }
}
impl Operand for F {
    fn id(self) -> u32 {
        constant(self)
    }
}

macro_rules! binary {
    ($trait:ident, $method:ident, $op:expr, $lhs:ty, $rhs:ty) => {
        impl $trait<$rhs> for $lhs {
            type Output = Expr;
            fn $method(self, rhs: $rhs) -> Expr {
                let (a, b) = (Operand::id(self), Operand::id(rhs));
                Expr(emit($op, a, b))
            }
        }
    };
}
macro_rules! arithmetic {
    ($lhs:ty, $rhs:ty) => {
        binary!(Add, add, ADD, $lhs, $rhs);
        binary!(Sub, sub, SUB, $lhs, $rhs);
        binary!(Mul, mul, MUL, $lhs, $rhs);
    };
}
arithmetic!(Expr, Expr);
arithmetic!(Expr, Var);
arithmetic!(Expr, F);
arithmetic!(Var, Var);
arithmetic!(Var, Expr);
arithmetic!(Var, F);

impl Neg for Expr {
    type Output = Expr;
    fn neg(self) -> Expr {
        Expr(emit(NEG, self.0, 0))
    }
}
impl Neg for Var {
    type Output = Expr;
    fn neg(self) -> Expr {
        Expr(emit(NEG, self.0, 0))
    }
}
impl AddAssign for Expr {
    fn add_assign(&mut self, rhs: Self) {
        *self = *self + rhs;
    }
}
impl SubAssign for Expr {
    fn sub_assign(&mut self, rhs: Self) {
        *self = *self - rhs;
    }
}
impl MulAssign for Expr {
    fn mul_assign(&mut self, rhs: Self) {
        *self = *self * rhs;
    }
}
impl Sum for Expr {
    fn sum<I: Iterator<Item = Self>>(iter: I) -> Self {
        iter.fold(Self::zero(), |acc, x| acc + x)
    }
}
impl Product for Expr {
    fn product<I: Iterator<Item = Self>>(iter: I) -> Self {
        iter.fold(Self::one(), |acc, x| acc * x)
    }
}

macro_rules! from_field_ctor {
    ($($name:ident($t:ty)),*) => { $(fn $name(n: $t) -> Self { Expr(constant(F::$name(n))) })* };
}
impl AbstractField for Expr {
    type F = F;
    fn zero() -> Self {
        Expr(constant(F::zero()))
    }
    fn one() -> Self {
        Expr(constant(F::one()))
    }
    fn two() -> Self {
        Expr(constant(F::two()))
    }
    fn neg_one() -> Self {
        Expr(constant(F::neg_one()))
    }
    fn from_f(f: F) -> Self {
        Expr(constant(f))
    }
    fn generator() -> Self {
        Expr(constant(F::generator()))
    }
    from_field_ctor!(
        from_bool(bool),
        from_canonical_u8(u8),
        from_canonical_u16(u16),
        from_canonical_u32(u32),
        from_canonical_u64(u64),
        from_canonical_usize(usize),
        from_wrapped_u32(u32),
        from_wrapped_u64(u64)
    );
}

/// Extension-field expressions: SP1's chips assert over the base field only (extension values belong to LogUp-GKR, which
/// is driven by `chip.sends()` / `receives()`, not by `eval`). The types exist because `ExtensionBuilder` is a supertrait
/// of the builder bound; every operation refuses, as the reference's builder does for `assert_zero_ext`.
#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash, Default)]
pub struct ExprEF;
#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
pub struct VarEF;

fn no_ef() -> ! {
    panic!("sp1-hip-export: extension-field constraint expressions are not part of the zerocheck constraint programs")
}
impl From<VarEF> for ExprEF {
    fn from(_: VarEF) -> Self {
        no_ef()
    }
}
impl From<EF> for ExprEF {
    fn from(_: EF) -> Self {
        no_ef()
    }
}
impl From<Expr> for ExprEF {
    fn from(_: Expr) -> Self {
        no_ef()
    }
}
macro_rules! ef_binary {
    ($($trait:ident $method:ident $assign:ident $assign_method:ident),*) => { $(
        impl $trait<ExprEF> for ExprEF { type Output = ExprEF; fn $method(self, _: ExprEF) -> ExprEF { no_ef() } }
        impl $trait<EF> for ExprEF { type Output = ExprEF; fn $method(self, _: EF) -> ExprEF { no_ef() } }
        impl $trait<VarEF> for ExprEF { type Output = ExprEF; fn $method(self, _: VarEF) -> ExprEF { no_ef() } }
        impl $trait<Expr> for ExprEF { type Output = ExprEF; fn $method(self, _: Expr) -> ExprEF { no_ef() } }
        impl $trait<VarEF> for VarEF { type Output = ExprEF; fn $method(self, _: VarEF) -> ExprEF { no_ef() } }
        impl $trait<ExprEF> for VarEF { type Output = ExprEF; fn $method(self, _: ExprEF) -> ExprEF { no_ef() } }
        impl $trait<EF> for VarEF { type Output = ExprEF; fn $method(self, _: EF) -> ExprEF { no_ef() } }
        impl $assign<ExprEF> for ExprEF { fn $assign_method(&mut self, _: ExprEF) { no_ef() } }
        impl $assign<Expr> for ExprEF { fn $assign_method(&mut self, _: Expr) { no_ef() } }
    )* };
}
ef_binary!(Add add AddAssign add_assign, Sub sub SubAssign sub_assign, Mul mul MulAssign mul_assign);
impl Neg for ExprEF {
    type Output = ExprEF;
    fn neg(self) -> ExprEF {
        no_ef()
    }
}
impl Sum for ExprEF {
    fn sum<I: Iterator<Item = Self>>(_: I) -> Self {
        no_ef()
    }
}
impl Product for ExprEF {
    fn product<I: Iterator<Item = Self>>(_: I) -> Self {
        no_ef()
    }
}
macro_rules! ef_ctor {
    ($($name:ident($t:ty)),*) => { $(fn $name(_: $t) -> Self { no_ef() })* };
}
impl AbstractField for ExprEF {
    type F = EF;
    fn zero() -> Self {
        ExprEF
    }
    fn one() -> Self {
        no_ef()
    }
    fn two() -> Self {
        no_ef()
    }
    fn neg_one() -> Self {
        no_ef()
    }
    fn from_f(_: EF) -> Self {
        no_ef()
    }
    fn generator() -> Self {
        no_ef()
    }
    ef_ctor!(
        from_bool(bool),
        from_canonical_u8(u8),
        from_canonical_u16(u16),
        from_canonical_u32(u32),
        from_canonical_u64(u64),
        from_canonical_usize(usize),
        from_wrapped_u32(u32),
        from_wrapped_u64(u64)
    );
}
impl AbstractExtensionField<Expr> for ExprEF {
    const D: usize = 4;
    fn from_base(_: Expr) -> Self {
        no_ef()
    }
    fn from_base_slice(_: &[Expr]) -> Self {
        no_ef()
    }
    fn from_base_fn<Func: FnMut(usize) -> Expr>(_: Func) -> Self {
        no_ef()
    }
    fn as_base_slice(&self) -> &[Expr] {
        no_ef()
    }
}

/// The builder a chip's `eval` runs against.
pub struct RecordingBuilder<'a> {
    preprocessed: RowMajorMatrixView<'a, Var>,
    main: RowMajorMatrixView<'a, Var>,
    public_values: &'a [Var],
}

impl<'a> AirBuilder for RecordingBuilder<'a> {
    type F = F;
    type Expr = Expr;
    type Var = Var;
    type M = RowMajorMatrixView<'a, Var>;

    fn main(&self) -> Self::M {
        self.main
    }
    fn is_first_row(&self) -> Expr {
        unimplemented!("single-row constraints only (crates/hypercube/src/folder.rs:L276-L323)")
    }
    fn is_last_row(&self) -> Expr {
        unimplemented!("single-row constraints only")
    }
    fn is_transition_window(&self, _: usize) -> Expr {
        unimplemented!("single-row constraints only")
    }
    fn assert_zero<I: Into<Expr>>(&mut self, x: I) {
        // the k-th assert of the tape is constraint k
        emit(ASSERT_ZERO, x.into().0, 0);
    }
}
impl ExtensionBuilder for RecordingBuilder<'_> {
    type EF = EF;
    type ExprEF = ExprEF;
    type VarEF = VarEF;
    fn assert_zero_ext<I: Into<ExprEF>>(&mut self, _: I) {
        no_ef()
    }
}
impl<'a> PermutationAirBuilder for RecordingBuilder<'a> {
    type MP = RowMajorMatrixView<'a, VarEF>;
    type RandomVar = VarEF;
    fn permutation(&self) -> Self::MP {
        unimplemented!("no permutation trace in SP1 Hypercube (lookups are LogUp-GKR)")
    }
    fn permutation_randomness(&self) -> &[VarEF] {
        unimplemented!()
    }
}
impl PairBuilder for RecordingBuilder<'_> {
    fn preprocessed(&self) -> Self::M {
        self.preprocessed
    }
}
impl AirBuilderWithPublicValues for RecordingBuilder<'_> {
    type PublicVar = Var;
    fn public_values(&self) -> &[Var] {
        self.public_values
    }
}
// lookups are not part of `eval`'s output here: the exporter reads them from `chip.sends()` / `chip.receives()`
impl EmptyMessageBuilder for RecordingBuilder<'_> {}
impl TrivialOperationBuilder for RecordingBuilder<'_> {}

/// Run `air.eval` over a fresh tape and return the SSA program.
pub fn record<A>(air: &A, preprocessed_width: usize, main_width: usize, num_public_values: usize) -> Vec<[u32; 3]>
where
    A: MachineAir<F> + for<'a> Air<RecordingBuilder<'a>>,
{
    TAPE.with(|t| *t.borrow_mut() = Tape::default());
    // column loads first, in column order (hash-consing makes later uses hit these values)
    let prep: Vec<Var> = (0..preprocessed_width as u32).map(|c| Var(emit(LOAD_PREP, c, 0))).collect();
    let main: Vec<Var> = (0..main_width as u32).map(|c| Var(emit(LOAD_MAIN, c, 0))).collect();
    let publics: Vec<Var> = (0..num_public_values as u32).map(|i| Var(emit(PUBLIC, i, 0))).collect();
    let prep_m = DenseMatrix::new(prep, preprocessed_width.max(1));
    let main_m = DenseMatrix::new(main, main_width.max(1));
    let mut builder = RecordingBuilder { preprocessed: prep_m.as_view(), main: main_m.as_view(), public_values: &publics };
    air.eval(&mut builder);
    // (values nobody uses — unused columns / public values — are harmless: the prover's planner only walks the cones of
    // the asserts)
    TAPE.with(|t| std::mem::take(&mut t.borrow_mut().instrs))
}
