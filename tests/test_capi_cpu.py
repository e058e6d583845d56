"""CPU tests of the C-ABI library: it loads, exports what include/sph_amd.h declares, its
host-side case set-up agrees with the oracle's restatement of InitSPH, and it refuses to
run without a GPU instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O
from sph_amd import capi

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sph_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libsph_amd.so does not export %s" % n
    assert set(names) == set(capi.SYMBOLS)


def test_struct_layouts_match_the_header():
    # sizes as a C compiler lays the header's structs out (natural alignment)
    assert C.sizeof(capi.Params) == 8 + 16 + 8 * 10 + 8 + 8 + 8 * 4 + 8 + 8 + 8 * 4 + 8 + 8
    assert C.sizeof(capi.Fields) == 8 * 16
    assert C.sizeof(capi.Scalars) == 8 + 8 + 8 * 4 + 8 * 4
    assert C.sizeof(capi.Case) == 16 + 8 + 24 + 24 + 8 + 32 + 8 + 8 + 24 + 8 + 24
    # sph_transport: user, host_buffers + struct_size in one 8-byte slot, three callbacks
    assert C.sizeof(capi.Transport) == 8 + 8 + 3 * 8
    assert capi.Transport.struct_size.offset == 12 and capi.Transport.sendrecv.offset == 16
    assert capi.make_transport(1, capi.SENDRECV_FN(), capi.ALLREDUCE_FN()).struct_size == 40


@pytest.mark.parametrize("program", [0, 1, 2])
def test_case_setup_matches_oracle(program):
    oc, pc = O.default_case(program), capi.default_case(program)
    p, simmass, boundmass = capi.case_params(pc)
    k = O.derive(oc)
    assert (p.dim, p.npts, p.bound_parts) == (k.dim, k.npts, k.bound_parts)
    for a, b in ((p.H, k.H), (p.HSQ, k.HSQ), (p.correc, k.correc), (p.search_radius_sq, k.search_radius),
                 (p.B, k.B), (p.gam, k.gam), (p.rho0, k.rho0), (p.Cs, k.Cs), (p.alpha, k.alpha),
                 (p.eps, k.eps), (p.g, k.g), (p.beta, k.beta), (simmass, k.simmass), (boundmass, k.boundmass)):
        assert a == b
    assert (p.g_axis, p.subits) == (k.g_axis, k.subits)
    _, st = capi.case_init(pc)
    ost = O.Oracle(oc).get_state(O.STATE_N)
    assert np.array_equal(st["x"].T, ost["x"])
    assert np.array_equal(st["m"], ost["m"]) and np.array_equal(st["rho"], ost["rho"])
    assert not st["v"].any() and not st["f"].any() and not st["p"].any()


def test_scaled_case_setup_matches_oracle():
    for name, spec in util.small_cases().items():
        oc, pc = util.both_cases(spec)
        _, st = capi.case_init(pc)
        ost = O.Oracle(oc).get_state(O.STATE_N)
        assert np.array_equal(st["x"].T, ost["x"]), name
        assert np.array_equal(st["m"], ost["m"]), name


def test_droplet_and_jet_setup_matches_oracle():
    for name, spec in util.drop_cases().items():
        oc, pc = util.both_cases(spec)
        p, st = capi.case_init(pc)
        orc = O.Oracle(oc)
        ost = orc.get_state(O.STATE_N)
        assert int(p.npts) == orc.npts and int(p.bound_parts) == orc.bound_parts, name
        assert np.array_equal(st["x"].T, ost["x"]) and np.array_equal(st["v"].T, ost["v"]), name
        assert np.array_equal(st["m"], ost["m"]), name
        plain = dict(spec)
        del plain["drop"]
        p0, _, _ = capi.case_params(util.both_cases(plain)[1])
        added = int(p.npts) - int(p0.npts)
        assert added > 20, name
        assert (p.H, p.mass) == (p0.H, p0.mass)  # the added body does not change the constants
        moving = np.any(st["v"] != 0.0, axis=0)
        assert moving.sum() == added and not moving[: int(p0.npts)].any(), name


def test_bad_case_is_rejected():
    c = capi.default_case(2)
    c.npart[1] = 0
    with pytest.raises(capi.SphError):
        capi.case_params(c)
    with pytest.raises(capi.SphError):
        capi.default_case(7)


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU refusal; a GPU is present")
def test_no_gpu_means_a_loud_error_not_a_fallback():
    p, _, _ = capi.case_params(capi.default_case(1))
    with pytest.raises(capi.SphError) as e:
        capi.Sim(p)
    assert "no HIP device" in str(e.value) or "failed" in str(e.value)


def test_product_code_does_not_reach_the_oracle():
    # the shipped package and the host driver must not import, link or call the checker
    for base in ("sph_amd",):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "libsph_amd.so"))
    with pytest.raises(capi.SphError) as e:
        capi.lib()
    assert "not built" in str(e.value) and "no CPU fallback" in str(e.value)


def test_create_rejects_bad_parameters_before_looking_for_a_device():
    import ctypes as C2

    def code_and_message(mutate):
        p, _, _ = capi.case_params(capi.default_case(2))
        mutate(p)
        h = C2.c_void_p()
        rc = capi.lib().sph_create(C2.byref(p), C2.byref(h))
        msg = capi.lib().sph_last_error(h).decode() if h else ""
        if h:
            capi.lib().sph_destroy(h)
        return rc, msg

    for mutate, text in ((lambda p: setattr(p, "dim", 4), "dim must be 2 or 3"),
                         (lambda p: setattr(p, "npts", 0), "npts out of range"),
                         (lambda p: setattr(p, "bound_parts", p.npts + 1), "bound_parts > npts"),
                         (lambda p: setattr(p, "variant", 0), "2-D only"),
                         (lambda p: setattr(p, "subits", 0), "bad physical constants"),
                         (lambda p: (setattr(p, "nranks", 2), setattr(p, "rank", 2)), "rank out of range"),
                         (lambda p: (setattr(p, "nranks", 2), setattr(p, "npts_global", p.npts),
                                     setattr(p, "capacity", p.npts)), "capacity must exceed npts"),
                         (lambda p: (setattr(p, "nranks", 2), setattr(p, "npts_global", p.npts),
                                     setattr(p, "capacity", 2 * p.npts), setattr(p, "slab_lo", 1.0),
                                     setattr(p, "slab_hi", 1.0)), "empty slab interval")):
        rc, msg = code_and_message(mutate)
        assert rc == capi.EINVAL and text in msg, (rc, msg, text)
