"""The GPU schedule (projective G2 steps + tracked scale, tower Fq12, cyclotomic hard part), stated
in big ints in tests/sched_model.py, must reproduce the reference restatement exactly."""
import random

import helpers as H
import sched_model as S
from helpers import R


def test_schedule_equals_reference():
    P, Q = H.subgroup_points(3, seed=21)
    rng = random.Random(2)
    x = [rng.randrange(R.P) for _ in range(12)]
    y = [rng.randrange(R.P) for _ in range(12)]
    assert S.to_list(S.fq12_mul(S.from_list(x), S.from_list(y))) == R.fq12_mul(x, y)
    assert S.to_list(S.fq12_sqr(S.from_list(x))) == R.fq12_mul(x, x)
    assert S.to_list(S.fq12_inv(S.from_list(x))) == R.fq12_inv(x)
    m = R.miller_loop_native(Q[0], P[0])
    assert S.miller_exact([(P[0], Q[0])]) == m
    assert S.miller_exact([(P[0], Q[0]), (P[1], Q[1]), (P[2], Q[2])]) == R.multi_miller_loop_native(list(zip(P, Q)))
    assert S.final_exp_gpu(x) == R.final_exp_native(x)
    assert S.final_exp_gpu(m) == R.final_exp_native(m)
    assert S.pairing_gpu(P[1], Q[1]) == R.pairing_myfq12(P[1], Q[1])


def test_cyclotomic_square_only_valid_in_subgroup():
    P, Q = H.subgroup_points(1, seed=22)
    e = S.from_list(R.pairing_myfq12(P[0], Q[0]))
    assert S.to_list(S.cyclotomic_sqr(e)) == S.to_list(S.fq12_sqr(e))
