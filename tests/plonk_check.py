"""The reference verifier (plonk/src/compiler/verifier.rs:187-217, plonk/src/proof_system/proof.rs:179-404) for
arithmetic-only circuits, in Python big integers (tests/model.py, oracle/pairing.py):
  verify()          the whole of it: transcript replay, the two pairing points, e(-A, [tau]_2) e(B, [1]_2) == 1;
  verifier_sides()  the same equation with the pairing replaced by the known trapdoor of a test SRS,
                    tau * A = B in G1, cheap enough for 2^20-gate proofs in every test run."""
import model as M

R = M.R_MOD
K = (1, 7, 13, 17)
COMM_NAMES = ["a_comm", "b_comm", "c_comm", "d_comm", "z_comm", "t_low_comm", "t_mid_comm", "t_high_comm",
              "t_fourth_comm", "w_z_chall_comm", "w_z_chall_w_comm"]
EVAL_NAMES = ["a_eval", "b_eval", "c_eval", "d_eval", "a_w_eval", "b_w_eval", "d_w_eval", "q_arith_eval", "q_c_eval",
              "q_l_eval", "q_r_eval", "s_sigma_1_eval", "s_sigma_2_eval", "s_sigma_3_eval", "z_eval"]
SEL_NAMES = ["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
             "q_variable_group_add", "s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"]


def opening_key_g2(tau):
    """(h, [tau] h) of the OpeningKey (kzg10/srs.rs:60-70) for a trapdoor given as an integer; h is the G2 generator
    taken from the reference's compressed vectors (entry 1 of g2_compressed_valid_test_vectors.dat)."""
    import os
    from conftest import GOLDEN
    from oracle import pairing as Pg
    data = open(os.path.join(GOLDEN, "g2_compressed_valid_test_vectors.dat"), "rb").read()
    h = Pg.g2_from_compressed(data[96:192])
    return h, Pg.g2_mul(h, tau)


def _ints(arr):
    import numpy as np
    return M.fr_vec_from_mont(np.asarray(arr, dtype=np.uint64).reshape(-1, 4))


def _pt(xy, inf=0):
    return M.affine_from_limbs(xy, inf)


def verifier_sides(n, proof, sel_xy, sel_inf, public_input0, tau, g_xy, u=None, tamper=0):
    """-> (left, right) of tau * (W_z + u W_zw) = z W_z + u z w W_zw + F - E; equal for a valid proof."""
    if u is None:
        u = M.random_fr_ints(1, 1234)[0]       # the verifier's own challenge; the equation must hold for any value
    a_pt, b_pt = pairing_points(n, proof, sel_xy, sel_inf, {0: public_input0}, g_xy, u, tamper=tamper)
    return M.ec_mul(a_pt, tau), b_pt


def verify(n, constraints, label, proof, sel_xy, sel_inf, public_inputs_mont, pi_indexes, g_xy, h_g2, tau_h_g2, orc):
    """Verifier::verify: True when the proof is accepted.  public_inputs_mont: (k, 4) Montgomery limbs in the order
    of pi_indexes; h_g2 / tau_h_g2: the OpeningKey's G2 points (kzg10/key.rs:165-186) as oracle.pairing tuples."""
    import numpy as np
    from oracle import pairing as Pg
    pis = np.asarray(public_inputs_mont, dtype=np.uint64).reshape(-1, 4)
    if pis.shape[0] != len(pi_indexes):
        return False                            # Error::InconsistentPublicInputsLen
    ch = orc.verifier_challenges(label, sel_xy, sel_inf, constraints, pis, proof["comms"], proof["comms_inf"], proof["evals"])
    replay = dict(proof)
    replay["challenges"] = ch[:10]              # the verifier trusts its own transcript, not the prover's report
    u = M.fr_from_mont(ch[10])
    pi = {idx: v for idx, v in zip(pi_indexes, _ints(pis))}
    a_pt, b_pt = pairing_points(n, replay, sel_xy, sel_inf, pi, g_xy, u)
    neg_a = None if a_pt is None else (a_pt[0], (-a_pt[1]) % M.P_MOD)
    return Pg.pairing_product_is_one([(neg_a, tau_h_g2), (b_pt, h_g2)])


def pairing_points(n, proof, sel_xy, sel_inf, public_inputs, g_xy, u, tamper=0):
    """-> (A, B) with A = W_z + u W_zw and B = z W_z + u z w W_zw + F - E (proof.rs:257-391); the proof is accepted
    when e(A, [tau]_2) = e(B, [1]_2).  public_inputs: {gate index: value}."""
    G = _pt(g_xy)
    comm = {name: _pt(proof["comms"][i], proof["comms_inf"][i]) for i, name in enumerate(COMM_NAMES)}
    e = dict(zip(EVAL_NAMES, _ints(proof["evals"])))
    beta, gamma, alpha, _, _, _, _, z, v, v_w = _ints(proof["challenges"])
    vk = {name: _pt(sel_xy[i], sel_inf[i]) for i, name in enumerate(SEL_NAMES)}
    om = M.domain_root(n)

    zh = (pow(z, n, R) - 1) % R
    l1 = zh * pow(n * (z - 1) % R, -1, R) % R
    # public input polynomial at z (barycentric form, proof.rs:511-561): sum of pi_j L_j(z)
    pi_eval = 0
    for idx, val in public_inputs.items():
        wj = pow(om, idx, R)
        pi_eval = (pi_eval + val * wj * zh % R * pow(n * (z - wj) % R, -1, R)) % R

    # linearization commitment (proof.rs:408-496, widget/arithmetic/verifierkey.rs:91-117,
    # widget/permutation/verifierkey.rs); the range/logic/ecc selectors of these circuits are identically zero
    qa = e["q_arith_eval"]
    terms = [(e["a_eval"] * e["b_eval"] * qa, vk["q_m"]), (e["a_eval"] * qa, vk["q_l"]), (e["b_eval"] * qa, vk["q_r"]),
             (e["c_eval"] * qa, vk["q_o"]), (e["d_eval"] * qa, vk["q_f"]), (qa, vk["q_c"])]
    _, _, _, range_ch, logic_ch, fixed_ch, var_ch = _ints(proof["challenges"])[:7]
    edwards_d = (-10240 * pow(10241, -1, R)) % R         # coset-jubjub/src/lib.rs:368-373
    if vk["q_fixed_group_add"] is not None:  # widget/ecc/scalar_mul/fixed_base/verifierkey.rs:45-99
        k1 = fixed_ch * fixed_ch % R
        acc_x, acc_x_w, acc_y, acc_y_w = e["a_eval"], e["a_w_eval"], e["b_eval"], e["b_w_eval"]
        xy_alpha = e["c_eval"]
        bit = (e["d_w_eval"] - 2 * e["d_eval"]) % R
        bit_consistency = bit * (bit - 1) * (bit + 1) % R
        y_alpha = (bit * bit * (e["q_r_eval"] - 1) + 1) % R
        x_alpha = e["q_l_eval"] * bit % R
        xy_cons = (bit * e["q_c_eval"] - xy_alpha) * k1 % R
        curve = xy_alpha * acc_x % R * acc_y % R * edwards_d % R
        x_cons = (acc_x_w + acc_x_w * curve - (x_alpha * acc_y + y_alpha * acc_x)) * k1 * k1 % R
        y_cons = (acc_y_w - acc_y_w * curve - (x_alpha * acc_x + y_alpha * acc_y)) * k1 ** 3 % R
        terms.append(((bit_consistency + x_cons + y_cons + xy_cons) * fixed_ch % R, vk["q_fixed_group_add"]))
    if vk["q_variable_group_add"] is not None:   # widget/ecc/curve_addition/verifierkey.rs:38-80
        k1 = var_ch * var_ch % R
        x1, x3, y1, y3, x2, y2, x1y2 = (e["a_eval"], e["a_w_eval"], e["b_eval"], e["b_w_eval"], e["c_eval"], e["d_eval"],
                                        e["d_w_eval"])
        y1x2 = y1 * x2 % R
        xy_cons = (x1 * y2 - x1y2) % R
        x3_cons = (x1y2 + y1x2 - (x3 + x3 * (edwards_d * x1y2 % R * y1x2 % R))) * k1 % R
        y3_cons = (y1 * y2 + x1 * x2 - (y3 - y3 * edwards_d % R * x1y2 % R * y1x2)) * k1 * k1 % R
        terms.append(((xy_cons + x3_cons + y3_cons) * var_ch % R, vk["q_variable_group_add"]))

    def delta(f):                           # range/proverkey.rs:92-97
        return f * (f - 1) * (f - 2) * (f - 3) % R

    if vk["q_range"] is not None:           # widget/range/verifierkey.rs:38-62
        k1 = range_ch * range_ch % R
        b_sum = (delta(e["c_eval"] - 4 * e["d_eval"]) + delta(e["b_eval"] - 4 * e["c_eval"]) * k1
                 + delta(e["a_eval"] - 4 * e["b_eval"]) * k1 * k1 + delta(e["d_w_eval"] - 4 * e["a_eval"]) * k1 ** 3) % R
        terms.append((b_sum * range_ch % R, vk["q_range"]))
    if vk["q_logic"] is not None:           # widget/logic/verifierkey.rs:38-84, logic/proverkey.rs:132-155
        k1 = logic_ch * logic_ch % R
        ai = (e["a_w_eval"] - 4 * e["a_eval"]) % R
        bi = (e["b_w_eval"] - 4 * e["b_eval"]) % R
        di = (e["d_w_eval"] - 4 * e["d_eval"]) % R
        w = e["c_eval"]
        f_term = w * (w * (4 * w - 18 * (ai + bi) + 81) + 18 * (ai * ai + bi * bi) - 81 * (ai + bi) + 83) % R
        xor_and = (e["q_c_eval"] * (9 * di - 3 * (ai + bi)) + 3 * (ai + bi + di) - 2 * f_term) % R
        c_sum = (delta(ai) + delta(bi) * k1 + delta(di) * k1 * k1 + (w - ai * bi) * k1 ** 3 + xor_and * k1 ** 4) % R
        terms.append((c_sum * logic_ch % R, vk["q_logic"]))
    ident = (e["a_eval"] + beta * z + gamma) * (e["b_eval"] + beta * K[1] * z + gamma) \
        * (e["c_eval"] + beta * K[2] * z + gamma) * (e["d_eval"] + beta * K[3] * z + gamma) * alpha % R
    terms.append(((ident + l1 * alpha * alpha + u) % R, comm["z_comm"]))
    copy = (e["a_eval"] + beta * e["s_sigma_1_eval"] + gamma) * (e["b_eval"] + beta * e["s_sigma_2_eval"] + gamma) \
        * (e["c_eval"] + beta * e["s_sigma_3_eval"] + gamma) * beta * e["z_eval"] * alpha % R
    terms.append(((-copy) % R, vk["s_sigma_4"]))
    mzh = (-zh) % R
    terms += [(mzh, comm["t_low_comm"]), (pow(z, n, R) * mzh % R, comm["t_mid_comm"]),
              (pow(z, 2 * n, R) * mzh % R, comm["t_high_comm"]), (pow(z, 3 * n, R) * mzh % R, comm["t_fourth_comm"])]
    lin = M.ec_msm([p for _, p in terms], [s for s, _ in terms])

    r0 = (pi_eval - l1 * alpha * alpha
          - alpha * (e["a_eval"] + beta * e["s_sigma_1_eval"] + gamma) * (e["b_eval"] + beta * e["s_sigma_2_eval"] + gamma)
          * (e["c_eval"] + beta * e["s_sigma_3_eval"] + gamma) * (e["d_eval"] + gamma) * e["z_eval"]) % R
    vs = [pow(v, i, R) for i in range(1, 8)]
    vs += [v_w * u % R, v_w * v_w * u % R, v_w * v_w * v_w * u % R]
    e_list = [e[k] for k in ("a_eval", "b_eval", "c_eval", "d_eval", "s_sigma_1_eval", "s_sigma_2_eval", "s_sigma_3_eval",
                             "a_w_eval", "b_w_eval", "d_w_eval")]
    e_scalar = (sum(x * y for x, y in zip(e_list, vs)) - r0 + u * e["z_eval"] + tamper) % R
    sc = vs[:7]
    sc[0] = (sc[0] + vs[7]) % R
    sc[1] = (sc[1] + vs[8]) % R
    sc[3] = (sc[3] + vs[9]) % R
    pts = [comm["a_comm"], comm["b_comm"], comm["c_comm"], comm["d_comm"], vk["s_sigma_1"], vk["s_sigma_2"], vk["s_sigma_3"]]
    F = M.ec_add(M.ec_msm(pts, sc), lin)
    E = M.ec_mul(G, e_scalar)
    wz, wzw = comm["w_z_chall_comm"], comm["w_z_chall_w_comm"]
    neg_e = None if E is None else (E[0], (-E[1]) % M.P_MOD)
    a_pt = M.ec_add(wz, M.ec_mul(wzw, u))
    b_pt = M.ec_add(M.ec_add(M.ec_mul(wz, z), M.ec_mul(wzw, u * z * om % R)), M.ec_add(F, neg_e))
    return a_pt, b_pt
