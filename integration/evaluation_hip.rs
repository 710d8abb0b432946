//! halo2_proofs/src/plonk/evaluation_hip.rs -- the `hip` bodies of `Evaluator::evaluate_h`
//! (plonk/evaluation.rs:1229-1241, the cuda signature) and of `evaluate` / `evaluate_with_theta`
//! (:2257-2398), over the plain-C descriptor of include/halo2_hip.h (`h2_evalh_desc`, mirrored as
//! `crate::hip::H2EvalhDesc`).
//!
//! UNCOMPILED in the build image (no Rust toolchain there); added to the crate by
//! integration/halo2_proofs_hip.patch.  It lives inside `crate::plonk` because it reads fields that are
//! `pub(in crate::plonk)` (the committed lookup / shuffle polynomials) and the private `evaluation` module.
//!
//! What the reference's cuda path does per proof: walk `ProveExpression` trees, one elementwise kernel
//! per node, re-deriving extended cosets through a 5-entry cache, then nine `eval_h_*` kernels.  Here the `Evaluator`'s own
//! straight-line program -- `constants`, `rotations`, `calculations`, `value_parts`, `lookup_results`,
//! `shuffle_results` (evaluation.rs:270-296), i.e. what the CPU twin interprets per row -- is handed to the library as
//! is, together with the coefficient forms, and ONE call returns h's numerator on the extended domain.  The library
//! turns the program into generated kernels itself the first time it sees it (csrc/evalh_gen.cpp: hipRTC, cached by program
//! hash in memory and on disk -- nothing to build or pass in from here; `hip::prepare_evaluate_h` moves that cost to keygen)
//! and deals the cosets of the extended domain over the `HALO2_PROOFS_N_GPU` device pool inside the call, as the cuda body
//! deals its gates / lookups / shuffles (evaluation.rs:1262-1275,1513-1520,1830-1837).
#![cfg(feature = "hip")]

use super::evaluation::{Calculation, Evaluator, LcChallenge, ValueSource};
use super::{logup, permutation, shuffle, Any, Expression, ProvingKey};
use crate::arithmetic::{CurveAffine, FieldExt};
use crate::hip::{
    self, H2Calculation, H2EvalhDesc, H2ValueSource, H2_ANY_ADVICE, H2_ANY_FIXED, H2_ANY_INSTANCE, H2_CALC_ADD,
    H2_CALC_ADD_CHALLENGE, H2_CALC_LC_CHALLENGE, H2_CALC_LC_THETA, H2_CALC_MUL, H2_CALC_NEGATE, H2_CALC_STORE,
    H2_CALC_SUB, H2_CHALLENGE_BETA, H2_CHALLENGE_GAMMA, H2_VS_ADVICE, H2_VS_CONSTANT, H2_VS_FIXED, H2_VS_INSTANCE,
    H2_VS_INTERMEDIATE,
};
use crate::poly::{Basis, Coeff, ExtendedLagrangeCoeff, Polynomial};
use std::marker::PhantomData;

/// `ValueSource` (evaluation.rs:44-57) -> `h2_value_source`
pub(in crate::plonk) fn flatten_value_source(v: &ValueSource) -> H2ValueSource {
    match *v {
        ValueSource::Constant(i) => H2ValueSource { kind: H2_VS_CONSTANT, index: i as u32, rot: 0 },
        ValueSource::Intermediate(i) => H2ValueSource { kind: H2_VS_INTERMEDIATE, index: i as u32, rot: 0 },
        ValueSource::Fixed(c, r) => H2ValueSource { kind: H2_VS_FIXED, index: c as u32, rot: r as u32 },
        ValueSource::Advice(c, r) => H2ValueSource { kind: H2_VS_ADVICE, index: c as u32, rot: r as u32 },
        ValueSource::Instance(c, r) => H2ValueSource { kind: H2_VS_INSTANCE, index: c as u32, rot: r as u32 },
    }
}

fn flatten_challenge(c: &LcChallenge) -> u32 {
    match c {
        LcChallenge::Beta => H2_CHALLENGE_BETA,
        LcChallenge::Gamma => H2_CHALLENGE_GAMMA,
    }
}

/// `Calculation` (evaluation.rs:95-112) -> `h2_calculation`
pub(in crate::plonk) fn flatten_calculation(c: &Calculation) -> H2Calculation {
    let none = H2ValueSource::default();
    let f = flatten_value_source;
    match c {
        Calculation::Add(a, b) => H2Calculation { op: H2_CALC_ADD, a: f(a), b: f(b), challenge: 0, power: 0 },
        Calculation::Sub(a, b) => H2Calculation { op: H2_CALC_SUB, a: f(a), b: f(b), challenge: 0, power: 0 },
        Calculation::Mul(a, b) => H2Calculation { op: H2_CALC_MUL, a: f(a), b: f(b), challenge: 0, power: 0 },
        Calculation::Negate(a) => H2Calculation { op: H2_CALC_NEGATE, a: f(a), b: none, challenge: 0, power: 0 },
        Calculation::LcChallenge(a, b, ch, p) => H2Calculation {
            op: H2_CALC_LC_CHALLENGE,
            a: f(a),
            b: f(b),
            challenge: flatten_challenge(ch),
            power: *p as u32,
        },
        Calculation::LcTheta(a, b) => H2Calculation { op: H2_CALC_LC_THETA, a: f(a), b: f(b), challenge: 0, power: 0 },
        Calculation::AddChallenge(a, ch) => H2Calculation {
            op: H2_CALC_ADD_CHALLENGE,
            a: f(a),
            b: none,
            challenge: flatten_challenge(ch),
            power: 0,
        },
        Calculation::Store(a) => H2Calculation { op: H2_CALC_STORE, a: f(a), b: none, challenge: 0, power: 0 },
    }
}

/// The program arrays of a descriptor, owned: the descriptor only borrows them.
pub(in crate::plonk) struct FlatProgram {
    pub rotations: Vec<i32>,
    pub calculations: Vec<H2Calculation>,
    pub value_parts: Vec<H2ValueSource>,
    pub lookup_sets: Vec<u32>,
    /// per lookup: table, product_0, sum_0, product_1, sum_1, ...
    pub lookup_calcs: Vec<H2Calculation>,
    /// per shuffle: input, shuffle
    pub shuffle_calcs: Vec<H2Calculation>,
}

impl FlatProgram {
    /// `Evaluator` (evaluation.rs:270-296) -> the arrays of `h2_evalh_desc`
    pub fn of<C: CurveAffine>(ev: &Evaluator<C>) -> Self {
        let mut lookup_sets = vec![];
        let mut lookup_calcs = vec![];
        for (table, products, sums) in ev.lookup_results.iter() {
            assert_eq!(products.len(), sums.len());
            lookup_sets.push(products.len() as u32);
            lookup_calcs.push(flatten_calculation(table));
            for (p, s) in products.iter().zip(sums.iter()) {
                lookup_calcs.push(flatten_calculation(p));
                lookup_calcs.push(flatten_calculation(s));
            }
        }
        let mut shuffle_calcs = vec![];
        for (input, shuffle) in ev.shuffle_results.iter() {
            shuffle_calcs.push(flatten_calculation(input));
            shuffle_calcs.push(flatten_calculation(shuffle));
        }
        FlatProgram {
            rotations: ev.rotations.clone(),
            calculations: ev.calculations.iter().map(|c| flatten_calculation(&c.calculation)).collect(),
            value_parts: ev.value_parts.iter().map(flatten_value_source).collect(),
            lookup_sets,
            lookup_calcs,
            shuffle_calcs,
        }
    }
}

fn ptrs<F, B>(polys: &[Polynomial<F, B>]) -> Vec<*const u64> {
    polys.iter().map(|p| p.values.as_ptr() as *const u64).collect()
}

impl<C: CurveAffine> Evaluator<C> {
    /// Evaluate h poly -- same signature and result as the cuda body (evaluation.rs:1229-1241): coefficient forms in,
    /// the numerator of h on the extended domain out.
    pub(in crate::plonk) fn evaluate_h(
        &self,
        pk: &ProvingKey<C>,
        advice_poly: Vec<&Vec<Polynomial<C::ScalarExt, Coeff>>>,
        instance_poly: Vec<&Vec<Polynomial<C::ScalarExt, Coeff>>>,
        y: C::ScalarExt,
        beta: C::ScalarExt,
        gamma: C::ScalarExt,
        theta: C::ScalarExt,
        lookups: &[Vec<logup::prover::Committed<C>>],
        shuffles: &[Vec<shuffle::prover::Committed<C>>],
        permutations: &[permutation::prover::Committed<C>],
    ) -> Polynomial<C::ScalarExt, ExtendedLagrangeCoeff> {
        // as the cuda path (evaluation.rs:1259): one circuit instance per proof
        assert!(advice_poly.len() == 1);
        let domain = &pk.vk.domain;
        let cs = &pk.vk.cs;
        let prog = FlatProgram::of(self);
        let constants = &self.constants;

        let fixed = ptrs(&pk.fixed_polys[..]);
        let advice = ptrs(&advice_poly[0][..]);
        let instance = ptrs(&instance_poly[0][..]);

        // permutation argument: z of every set, the permuted columns' (type, index), sigma polys (evaluation.rs:1017-1084)
        let sets = &permutations[0].sets;
        let perm_z: Vec<*const u64> =
            sets.iter().map(|s| s.permutation_product_poly.values.as_ptr() as *const u64).collect();
        let p = &cs.permutation;
        let perm_col_type: Vec<u32> = p
            .columns
            .iter()
            .map(|c| match c.column_type() {
                Any::Advice => H2_ANY_ADVICE,
                Any::Fixed => H2_ANY_FIXED,
                Any::Instance => H2_ANY_INSTANCE,
            })
            .collect();
        let perm_col_index: Vec<u32> = p.columns.iter().map(|c| c.index() as u32).collect();
        let perm_sigma = ptrs(&pk.permutation.polys[..]);

        // logup lookups: grand-sum polys of every set of every lookup (in order), multiplicity poly per lookup
        // (evaluation.rs:1138-1182); shuffles: product poly per shuffle (:1197-1219)
        let mut lookup_z = vec![];
        let mut lookup_m = vec![];
        for lookup in lookups[0].iter() {
            lookup_m.push(lookup.multiplicity_poly.values.as_ptr() as *const u64);
            for z in lookup.z_poly_set.iter() {
                lookup_z.push(z.values.as_ptr() as *const u64);
            }
        }
        assert_eq!(lookup_m.len(), prog.lookup_sets.len());
        assert_eq!(lookup_z.len(), prog.lookup_sets.iter().sum::<u32>() as usize);
        let shuffle_z: Vec<*const u64> =
            shuffles[0].iter().map(|s| s.product_poly.values.as_ptr() as *const u64).collect();
        assert_eq!(2 * shuffle_z.len(), prog.shuffle_calcs.len());

        let desc = H2EvalhDesc {
            k: domain.k(),
            extended_k: domain.extended_k(),
            blinding_factors: cs.blinding_factors() as u32,
            chunk_len: (cs.degree() - 2) as u32,
            constants: constants.as_ptr() as *const u64,
            n_constants: constants.len() as u32,
            rotations: prog.rotations.as_ptr(),
            n_rotations: prog.rotations.len() as u32,
            calculations: prog.calculations.as_ptr(),
            n_calculations: prog.calculations.len() as u32,
            value_parts: prog.value_parts.as_ptr(),
            n_value_parts: prog.value_parts.len() as u32,
            n_lookups: prog.lookup_sets.len() as u32,
            lookup_sets: prog.lookup_sets.as_ptr(),
            lookup_calcs: prog.lookup_calcs.as_ptr(),
            n_shuffles: shuffle_z.len() as u32,
            shuffle_calcs: prog.shuffle_calcs.as_ptr(),
            fixed: fixed.as_ptr(),
            n_fixed: fixed.len() as u32,
            advice: advice.as_ptr(),
            n_advice: advice.len() as u32,
            instance: instance.as_ptr(),
            n_instance: instance.len() as u32,
            // with `hip` the proving key has the cuda shape (plonk.rs:226-240): l0 / l_last in coefficient form,
            // l_active_row on the extended domain
            l0: pk.l0.values.as_ptr() as *const u64,
            l_last: pk.l_last.values.as_ptr() as *const u64,
            l_active_row: pk.l_active_row.values.as_ptr() as *const u64,
            n_perm_sets: perm_z.len() as u32,
            perm_z: perm_z.as_ptr(),
            n_perm_columns: perm_col_type.len() as u32,
            perm_col_type: perm_col_type.as_ptr(),
            perm_col_index: perm_col_index.as_ptr(),
            perm_sigma: perm_sigma.as_ptr(),
            lookup_z: lookup_z.as_ptr(),
            lookup_m: lookup_m.as_ptr(),
            shuffle_z: shuffle_z.as_ptr(),
            y: hip::limbs(&y),
            beta: hip::limbs(&beta),
            gamma: hip::limbs(&gamma),
            theta: hip::limbs(&theta),
            delta: hip::limbs(&C::Scalar::DELTA),
            zeta: hip::limbs(&C::Scalar::ZETA),
            extended_omega: hip::limbs(&domain.get_extended_omega()),
            reserved: std::ptr::null(),
            flags: 0,
            row_begin: 0,
            row_count: 0,
        };
        let values: Vec<C::ScalarExt> = hip::evaluate_h(&desc, true);
        Polynomial { values, _marker: PhantomData }
    }
}

/// A tiny `Evaluator`-style flattener for the expression lists of `evaluate_with_theta` (generic over the field, not
/// over a curve): the same rules as `Evaluator::add_expression` (evaluation.rs:661-775) without its sharing of common
/// sub-expressions, which a handful of lookup / shuffle expressions does not need.
struct LcProgram<F> {
    constants: Vec<F>,
    rotations: Vec<i32>,
    calculations: Vec<H2Calculation>,
}

impl<F: FieldExt> LcProgram<F> {
    fn new() -> Self {
        LcProgram { constants: vec![F::zero(), F::one()], rotations: vec![], calculations: vec![] }
    }

    fn constant(&mut self, c: F) -> H2ValueSource {
        let index = match self.constants.iter().position(|x| *x == c) {
            Some(i) => i,
            None => {
                self.constants.push(c);
                self.constants.len() - 1
            }
        };
        H2ValueSource { kind: H2_VS_CONSTANT, index: index as u32, rot: 0 }
    }

    fn rotation(&mut self, r: i32) -> u32 {
        match self.rotations.iter().position(|x| *x == r) {
            Some(i) => i as u32,
            None => {
                self.rotations.push(r);
                (self.rotations.len() - 1) as u32
            }
        }
    }

    fn push(&mut self, op: u32, a: H2ValueSource, b: H2ValueSource) -> H2ValueSource {
        self.calculations.push(H2Calculation { op, a, b, challenge: 0, power: 0 });
        H2ValueSource { kind: H2_VS_INTERMEDIATE, index: (self.calculations.len() - 1) as u32, rot: 0 }
    }

    fn add(&mut self, expr: &Expression<F>) -> H2ValueSource {
        let none = H2ValueSource::default();
        match expr {
            Expression::Constant(c) => self.constant(*c),
            Expression::Selector(_) => panic!("virtual selectors are removed during optimization"),
            Expression::Fixed { column_index, rotation, .. } => {
                let rot = self.rotation(rotation.0);
                H2ValueSource { kind: H2_VS_FIXED, index: *column_index as u32, rot }
            }
            Expression::Advice { column_index, rotation, .. } => {
                let rot = self.rotation(rotation.0);
                H2ValueSource { kind: H2_VS_ADVICE, index: *column_index as u32, rot }
            }
            Expression::Instance { column_index, rotation, .. } => {
                let rot = self.rotation(rotation.0);
                H2ValueSource { kind: H2_VS_INSTANCE, index: *column_index as u32, rot }
            }
            Expression::Negated(a) => {
                let a = self.add(a);
                self.push(H2_CALC_NEGATE, a, none)
            }
            Expression::Sum(a, b) => {
                let a = self.add(a);
                let b = self.add(b);
                self.push(H2_CALC_ADD, a, b)
            }
            Expression::Product(a, b) => {
                let a = self.add(a);
                let b = self.add(b);
                self.push(H2_CALC_MUL, a, b)
            }
            Expression::Scaled(a, f) => {
                let a = self.add(a);
                let f = self.constant(*f);
                self.push(H2_CALC_MUL, a, f)
            }
        }
    }
}

/// `evaluate_with_theta` (evaluation.rs:2330-2398; `evaluate` is the one-expression case): fold(0, acc * theta + e_i)
/// over the `size`-point domain of the given columns -- the evaluator program with `y := theta` and
/// `extended_k := k` (value_parts are Horner-folded in y), through the extended-coset entry point on host columns.
/// The pure-column fast paths of the reference (:2266-2276) stay with the caller.
pub(in crate::plonk) fn evaluate_lc<F: FieldExt, B: Basis>(
    expressions: &[Expression<F>],
    size: usize,
    rot_scale: i32,
    fixed: &[Polynomial<F, B>],
    advice: &[Polynomial<F, B>],
    instance: &[Polynomial<F, B>],
    theta: F,
) -> Vec<F> {
    assert!(size.is_power_of_two());
    // every caller passes rot_scale = 1 (the n-point Lagrange domain: logup/prover.rs:86-96, shuffle/prover.rs:73-83)
    assert_eq!(rot_scale, 1, "evaluate_with_theta over an extended domain is not routed to the GPU");
    let mut prog = LcProgram::<F>::new();
    let value_parts: Vec<H2ValueSource> = expressions.iter().map(|e| prog.add(e)).collect();
    let k = size.trailing_zeros();
    // the generator of the size-point domain, as EvaluationDomain::new derives it (poly/domain.rs:61-81); the program has
    // no permutation part, so it is carried for completeness only
    let mut omega = F::root_of_unity();
    for _ in k..F::S {
        omega = omega.square();
    }
    let fixed_p = ptrs(fixed);
    let advice_p = ptrs(advice);
    let instance_p = ptrs(instance);
    let zero = [0u64; 4];
    let desc = H2EvalhDesc {
        k,
        extended_k: k,
        blinding_factors: 0,
        chunk_len: 1,
        constants: prog.constants.as_ptr() as *const u64,
        n_constants: prog.constants.len() as u32,
        rotations: prog.rotations.as_ptr(),
        n_rotations: prog.rotations.len() as u32,
        calculations: prog.calculations.as_ptr(),
        n_calculations: prog.calculations.len() as u32,
        value_parts: value_parts.as_ptr(),
        n_value_parts: value_parts.len() as u32,
        n_lookups: 0,
        lookup_sets: std::ptr::null(),
        lookup_calcs: std::ptr::null(),
        n_shuffles: 0,
        shuffle_calcs: std::ptr::null(),
        fixed: fixed_p.as_ptr(),
        n_fixed: fixed_p.len() as u32,
        advice: advice_p.as_ptr(),
        n_advice: advice_p.len() as u32,
        instance: instance_p.as_ptr(),
        n_instance: instance_p.len() as u32,
        l0: std::ptr::null(),
        l_last: std::ptr::null(),
        l_active_row: std::ptr::null(),
        n_perm_sets: 0,
        perm_z: std::ptr::null(),
        n_perm_columns: 0,
        perm_col_type: std::ptr::null(),
        perm_col_index: std::ptr::null(),
        perm_sigma: std::ptr::null(),
        lookup_z: std::ptr::null(),
        lookup_m: std::ptr::null(),
        shuffle_z: std::ptr::null(),
        y: hip::limbs(&theta),
        beta: zero,
        gamma: zero,
        theta: hip::limbs(&theta),
        delta: hip::limbs(&F::DELTA),
        zeta: hip::limbs(&F::ZETA),
        extended_omega: hip::limbs(&omega),
        reserved: std::ptr::null(),
        flags: 0,
        row_begin: 0,
        row_count: 0,
    };
    hip::evaluate_h(&desc, false)
}
