"""fast_amd/fitsio.py against the FITS standard 4.0 (the reference writes its result files with astropy, fast/fast.py:809-812,
998-1002; astropy is not installed here).  Two independent checks of a written file: (1) byte-for-byte equality with a file
whose cards are TYPED OUT below from the standard's fixed-format rules (and with the committed copy of those bytes,
tests/golden/fits_three_values.fits); (2) a validator written from the standard's text, not from the writer."""
import os
import re
import struct

import numpy as np
import pytest

from fast_amd import fitsio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fits_three_values.fits")

#        1234567890123456789012345678901234567890
CARDS = ["SIMPLE  =                    T",
         "BITPIX  =                  -64",
         "NAXIS   =                    1",
         "NAXIS1  =                    3",
         "EXTEND  =                    T",
         "ZENITH  =                 55.0",
         "NITER   =                    3",
         "AO_MODE = 'NOAO    '",
         "OTRSCALE= 'inf     '",
         "ALIAS   = 'True    '",
         "DIFFLIM = 3.111218108226167E-06",
         "NOTE    = 'it''s a quote'",
         "END"]
VALUES = [1.0, -2.5, 3.111218108226167e-06]


def expected_bytes():
    hdr = "".join(c.ljust(80) for c in CARDS).encode("ascii")
    hdr += b" " * (-len(hdr) % 2880)
    body = b"".join(struct.pack(">d", v) for v in VALUES)
    return hdr + body + b"\0" * (-len(body) % 2880)


def validate(buf):
    """FITS 4.0: 3.1 (blocks), 4.1 (card layout), 4.2 (value formats), 4.4.1 (mandatory keywords), 5.3 (IEEE data)."""
    assert len(buf) % 2880 == 0 and len(buf) >= 2 * 2880
    cards, pos, end = [], 0, False
    while not end:
        block = buf[pos:pos + 2880]
        assert len(block) == 2880 and all(32 <= b <= 126 for b in block), "header: printable ASCII only"
        pos += 2880
        for i in range(0, 2880, 80):
            card = block[i:i + 80].decode("ascii")
            if end:
                assert card == " " * 80, "only blanks after END"
                continue
            if card[:8] == "END     ":
                assert card[8:] == " " * 72
                end = True
                continue
            cards.append(card)
    keys = [c[:8].rstrip() for c in cards]
    assert keys[:3] == ["SIMPLE", "BITPIX", "NAXIS"]
    vals = {}
    for c in cards:
        key = c[:8]
        assert re.match(r"^[A-Z0-9_-]+ *$", key), key
        assert c[8:10] == "= "
        v = c[10:]
        if v.lstrip().startswith("'"):
            assert v.startswith("'"), "fixed format: the string starts in column 11"
            m = re.match(r"^'((?:[^']|'')*)' *(/.*)?$", v)
            assert m, c
            assert len(m.group(1)) >= 8, "fixed format: at least 8 characters between the quotes"
            vals[key.rstrip()] = m.group(1).replace("''", "'").rstrip()
        else:
            tok = v.split("/")[0]
            field = tok.rstrip()
            if field.strip() in ("T", "F"):
                assert len(field) == 20 and field[19] in "TF", "logical in column 30"
                vals[key.rstrip()] = field.strip() == "T"
            elif re.match(r"^ *[+-]?\d+$", field):
                assert len(field) == 20, "integer right-justified to column 30"
                vals[key.rstrip()] = int(field)
            else:
                assert re.match(r"^ *[+-]?(\d+\.?\d*|\.\d+)([ED][+-]?\d+)?$", field), c
                assert ("." in field) or ("E" in field) or ("D" in field)
                assert len(field) >= 20 and (len(field) == 20 or not field.startswith(" ")), "real right-justified to column 30 (or longer, free format)"
                vals[key.rstrip()] = float(field.replace("D", "E"))
    assert vals["SIMPLE"] is True and vals["BITPIX"] == -64
    nax = vals["NAXIS"]
    assert keys[3:3 + nax] == [f"NAXIS{i + 1}" for i in range(nax)]
    n = int(np.prod([vals[f"NAXIS{i + 1}"] for i in range(nax)])) if nax else 0
    data = buf[pos:]
    assert len(data) == -(-8 * n // 2880) * 2880
    assert data[8 * n:] == b"\0" * (len(data) - 8 * n), "data padded with zero bytes"
    return vals, np.frombuffer(data[:8 * n], dtype=">f8").astype(float)


def test_written_file_is_the_typed_out_fits_file(tmp_path):
    f = tmp_path / "three.fits"
    fitsio.writeto(str(f), np.array(VALUES), header={"ZENITH": 55.0, "NITER": 3, "AO_MODE": "NOAO", "OTRSCALE": str(np.inf), "ALIAS": str(True),
                                                     "DIFFLIM": 3.111218108226167e-06, "NOTE": "it's a quote"})
    got = f.read_bytes()
    assert got == expected_bytes()
    assert got == open(GOLDEN, "rb").read()
    vals, data = validate(got)
    assert vals["ZENITH"] == 55.0 and vals["NITER"] == 3 and vals["AO_MODE"] == "NOAO" and vals["NOTE"] == "it's a quote"
    np.testing.assert_array_equal(data, VALUES)
    hdr, back = fitsio.read(str(f))
    np.testing.assert_array_equal(back, VALUES)
    assert hdr["DIFFLIM"] == 3.111218108226167e-06 and hdr["OTRSCALE"] == "inf"


def test_result_file_of_a_run_passes_the_validator(tmp_path):
    """The header Fast.save writes (make_header, fast/fast.py:771-807: every key and value type it uses) on a 2-D array."""
    f = tmp_path / "res.fits"
    hdr = {"ZENITH": 55, "WVL": 1550, "OTRSCALE": "inf", "INRSCALE": 1e-6, "POWER": 1.0, "PAA": 0.0, "AO_MODE": "AO", "TLOOP": 0.001,
           "TEXP": 0.001, "DSUBAP": 0.1, "ALIAS": "True", "NOISE": 0, "D_GND": 0.8, "OBSC_GND": 0, "D_SAT": 0.1, "OBSC_SAT": 0, "AXICON": "False",
           "W0": 0.3567852394658434, "L_SAT": np.float64(41127657.3), "H_SAT": 36e6, "DX": 0.01, "NPXLS": np.int64(1024), "NITER": 10000,
           "R0": 0.0712, "THETA0": 1.1e-05, "TAU0": 0.0021, "DIFFLIM": 3.111218108226167e-06, "SEED": 1, "FLAG": np.bool_(True)}
    data = np.random.default_rng(0).normal(size=(5, 400))
    fitsio.writeto(str(f), data, header=hdr)
    vals, flat = validate(f.read_bytes())
    assert vals["NAXIS"] == 2 and vals["NAXIS1"] == 400 and vals["NAXIS2"] == 5 and vals["FLAG"] is True
    np.testing.assert_array_equal(flat.reshape(5, 400), data)
    for k, v in hdr.items():
        assert vals[k] == (v if not isinstance(v, (np.generic,)) else v.item())


def test_what_does_not_fit_a_card_is_an_error(tmp_path):
    for bad in ({"TOOLONGKEY": 1}, {"lower case": 1}, {"NOTE": "x" * 69}, {"NOTE": "café"}):
        with pytest.raises(ValueError):
            fitsio.writeto(str(tmp_path / "bad.fits"), np.zeros(1), header=bad, overwrite=True)
