"""GPU suite: the HIP path (through the C-ABI) against the golden vectors captured from the
unmodified reference (tests/golden/): encoder, decode_scl_llr on seeded inputs regenerated ON THE
DEVICE (sha256-pinned), and the edge-case inputs (zeros, +-huge > 709.78, ties, |llr| = 40)."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu
libc = C.CDLL(None)


def _gpu_code(name):
    import polar_amd
    c, frozen, order, crcm = G.tables(name)
    libc.srand(1)
    if c.get("explicit_tables"):
        g = polar_amd.PolarCode.from_tables(c["n"], c["K"], c["crc"], frozen, order, None)
    else:
        g = polar_amd.PolarCode(c["n"], c["K"], c["eps"], c["crc"])
    assert (g.channel_order_descending == order).all() and (g.crc_matrix == crcm).all()
    return c, g


@pytest.mark.parametrize("name", G.code_names())
def test_encode_golden(built_lib, name):
    c, g = _gpu_code(name)
    info, coded = G.enc_vectors(name)
    assert (g.encode(info) == coded).all()
    assert (g.encode(info[3]) == coded[3]).all()


@pytest.mark.parametrize("name,ci", G.all_case_ids())
def test_decode_scl_llr_golden(built_lib, name, ci):
    import torch
    c, g = _gpu_code(name)
    cs, want = list(G.cases(name))[ci]
    B, N, K = cs["B"], 1 << c["n"], c["K"]
    d_llr = torch.empty((B, N), dtype=torch.float64, device="cuda")
    d_out = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    if "constellation" in cs:      # config 5: 16-ASK Gray BICM front end on the device
        g.synth_bicm_llr_dev(cs["constellation"], G.seed(), cs["trial0"], B, cs["snr_db"], d_llr.data_ptr())
    else:
        s = float.fromhex(cs["s_hex"])
        assert g.snr_sqrt_linear(cs["ebno"]) == s
        g.synth_llr_dev(G.seed(), cs["trial0"], B, s, d_llr.data_ptr())
    g.decode_scl_llr_dev(d_llr.data_ptr(), B, cs["L"], d_out.data_ptr())
    torch.cuda.synchronize()
    llr = d_llr.cpu().numpy()
    assert G.sha(llr) == cs["llr_sha256"], "device synth differs from the golden inputs (bitwise)"
    got = d_out.cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size}/{B} codewords differ from the reference, first {bad[:5]}"
    # host-pointer entry point gives the same answer
    assert (g.decode_scl_llr(llr[:8], cs["L"]) == want[:8]).all()


@pytest.mark.parametrize("name", [n for n in G.code_names() if G.load()[1]["codes"][n]["specials"]])
def test_decode_edge_inputs_golden(built_lib, name):
    c, g = _gpu_code(name)
    for sname, llr, exp in G.specials(name):
        for L, want in exp.items():
            got = g.decode_scl_llr(llr, L)
            assert (got == want).all(), f"{name}/{sname} L={L}"
