"""Validate the C restatement directly against the reference itself on seeded inputs.
Runs only where oracle/_ref/libff_ref.so exists (this container; it travels to the GPU
box as a prebuilt .so, the reference sources do not).  CPU only."""
import numpy as np
import pytest

from common import GROUPS


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_generators_and_primitives(port, ref, name, curve, group):
    assert port.sizes(curve, group) == ref.sizes(curve, group)
    assert (port.scalars_sha512(curve, 12345, 40) == ref.scalars_sha512(curve, 12345, 40)).all()
    assert (port.bases_seq(curve, group, 33, first=77) == ref.bases_seq(curve, group, 33, first=77)).all()
    assert (port.bases_r32(curve, group, 64) == ref.bases_r32(curve, group, 64)).all()
    pts = ref.bases_seq(curve, group, 8, first=5)
    sc = ref.scalars_sha512(curve, 3, 8)
    for i in range(8):
        a = ref.scalar_mul(curve, group, pts[i], sc[i])
        assert (port.scalar_mul(curve, group, pts[i], sc[i]) == a).all()
        b = ref.group_op(curve, group, 2, a)
        for op, second in ((0, b), (1, pts[(i + 1) % 8]), (2, None), (3, None), (4, None), (5, b)):
            assert (port.group_op(curve, group, op, a, second) == ref.group_op(curve, group, op, a, second)).all()
        assert port.group_op(curve, group, 6, a, a) == ref.group_op(curve, group, 6, a, a) == 1
        assert port.group_op(curve, group, 6, a, b) == ref.group_op(curve, group, 6, a, b) == 0
    plain = ref.fr_as_bigint(curve, sc)
    assert (port.fr_as_bigint(curve, sc) == plain).all()
    assert (port.fr_from_bigint(curve, plain) == sc).all()
    for c in (2, 7, 13, 18):
        for k in range(0, 30, 3):
            assert port.signed_digit(curve, plain[0], c, k) == ref.signed_digit(curve, plain[0], c, k)
            assert port.digit(curve, plain[1], c, k) == ref.digit(curve, plain[1], c, k)


@pytest.mark.parametrize("name,curve,group", GROUPS)
@pytest.mark.parametrize("n", [1, 7, 100, 513])
def test_multi_exp_matches_reference(port, ref, name, curve, group, n):
    if n == 513 and (curve == 2 or group == 2):
        n = 130   # keep the wide-field cases inside the CPU-suite budget
    bases = ref.bases_seq(curve, group, n, first=9)
    scalars = ref.scalars_sha512(curve, 1000, n)
    want = ref.multi_exp(curve, group, bases, scalars, ref.BDLO12_SIGNED, ref.FORM_SPECIAL)
    for method in (port.BDLO12_SIGNED, port.BDLO12):
        for form in (0, 1):
            for chunks in (1, 3):
                assert (port.multi_exp(curve, group, bases, scalars, method, form, chunks) == want).all()
    r32 = ref.bases_r32(curve, group, n)
    want = ref.multi_exp(curve, group, r32, scalars, ref.BDLO12_SIGNED, ref.FORM_SPECIAL)
    assert (port.multi_exp(curve, group, r32, scalars, port.BDLO12_SIGNED, 1) == want).all()


def test_window_heuristics_match(port, ref):
    for n in list(range(1, 70)) + [255, 256, 257, 1000, 4096, 1 << 16, (1 << 20) + 1, 1 << 26]:
        assert port.bdlo12_signed_optimal_c(n) == ref.bdlo12_signed_optimal_c(n)
        assert port.pippenger_optimal_c(n) == ref.pippenger_optimal_c(n)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_disk_format_and_stream(port, ref, name, curve, group):
    n = 70
    b = ref.bases_seq(curve, group, n, first=2)
    b[3] = ref.group_consts(curve, group)[1]
    b[5] = ref.group_op(curve, group, 2, b[5])
    disk = ref.disk_write(curve, group, b)
    assert (port.disk_write(curve, group, b) == disk).all()
    # (ref.multi_exp_stream is not called here: the reference's streaming reader crashes
    # intermittently in this build; the byte format above is what pins the stream path)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_precompute_table_and_msm(port, ref, name, curve, group):
    """[2^(kc)]P tables and multi_exp_stream_with_precompute, restatement vs the reference, with a
    zero base and a non-affine base in the input."""
    n, c = 40, 7
    bits = ref.sizes(curve, group)["fr_bits"]
    D = (bits + c - 1) // c
    b = ref.bases_seq(curve, group, n, first=9)
    b[4] = ref.group_consts(curve, group)[1]
    b[6] = ref.group_op(curve, group, 2, b[6])
    sc = ref.scalars_sha512(curve, 300, n)
    tab = ref.precompute_table(curve, group, b, c, D)
    assert (port.precompute_table(curve, group, b, c) == tab).all()
    want = ref.multi_exp_stream_with_precompute(curve, group, ref.disk_write(curve, group, tab), sc, c)
    assert (port.multi_exp_precompute(curve, group, tab, sc, c) == want).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_batch_exp_matches_reference(port, ref, name, curve, group):
    bits = ref.sizes(curve, group)["fr_bits"]
    g = ref.group_op(curve, group, 2, ref.bases_seq(curve, group, 1, first=6)[0])
    v = ref.scalars_sha512(curve, 11, 9)
    for w in (1, 4, 6):
        assert (port.batch_exp(curve, group, bits, w, g, v) == ref.batch_exp(curve, group, bits, w, g, v)).all()
    assert (port.batch_exp(curve, group, bits, 3, g, v, coeff=v[2]) ==
            ref.batch_exp(curve, group, bits, 3, g, v, coeff=v[2])).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_compressed_codec_vs_reference(port, ref, name, curve, group):
    """Compressed records of 24 further curve points (other x seeds than the fixtures) and of
    multiples of the generator: restatement == reference, both directions."""
    pts, _ = ref.curve_points(curve, group, 1000 + 17 * curve + group, 24)
    pts = np.concatenate([pts, ref.bases_seq(curve, group, 8, first=5)])
    enc = ref.disk_write_compressed(curve, group, pts)
    assert (port.disk_write_compressed(curve, group, pts) == enc).all()
    back, bad = port.disk_read_compressed(curve, group, enc, pts.shape[0])
    assert bad == 0 and (back == ref.disk_read_compressed(curve, group, enc, pts.shape[0])).all()
    assert (back == pts).all()
