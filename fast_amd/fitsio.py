"""Minimal FITS primary-HDU writer / reader (float64 image + scalar header cards).

The reference saves results with astropy (`fits.writeto`, fast/fast.py:809-812) and reloads them
with `fits.getheader` / `fits.getdata` (fast.py:998-1002); astropy is not installed here, so this
module writes the same standard file by hand, to the rules of the FITS standard 4.0 (sections 3.1, 4.1, 4.2, 4.4.1, 5.3):
2880-byte blocks; 80-character ASCII cards, keyword in columns 1-8 (upper case, digits, `_`, `-`), `= ` in columns 9-10;
fixed format -- logical T / F in column 30, integers and reals right-justified to column 30, strings from column 11 in
single quotes, at least eight characters, quotes doubled; mandatory SIMPLE, BITPIX, NAXIS, NAXISn first and in that order, END
last, header padded with blanks; big-endian IEEE-754 BITPIX = -64 data padded with zero bytes.  tests/test_fits_conformance.py
checks a written file byte for byte against cards typed out from those rules and runs an independent validator over it; astropy
is not available here, so reading by astropy itself is not tested."""
import os
import re

import numpy as np

BLOCK = 2880
_KEY = re.compile(r"^[A-Z0-9_-]{1,8}$")


def _card(key, value):
    key = str(key).upper()
    if not _KEY.match(key):
        raise ValueError(f"FITS keyword {key!r}: 1-8 characters of A-Z, 0-9, '_' and '-' (standard 4.1.2.1)")
    if isinstance(value, (bool, np.bool_)):
        v = f"{'T' if value else 'F':>20}"
    elif isinstance(value, (int, np.integer)):
        v = f"{int(value):>20d}"
    elif isinstance(value, (float, np.floating)):
        # a real needs a decimal point or an exponent (4.2.4); NaN / infinity have no FITS representation: written as strings
        v = f"{repr(float(value)).upper():>20}" if np.isfinite(value) else f"'{str(float(value)):<8}'"
    else:
        s = str(value)
        if not all(32 <= ord(c) <= 126 for c in s):
            raise ValueError(f"FITS string value of {key}: printable ASCII only (standard 4.2.1)")
        s = s.replace("'", "''")
        if len(s) > 68:
            raise ValueError(f"FITS string value of {key} does not fit one card (68 characters)")
        v = f"'{s:<8}'"
    card = f"{key:<8}= {v}"
    if len(card) > 80:
        raise ValueError(f"FITS card of {key} exceeds 80 characters")
    return card.ljust(80)


def writeto(fname, data, header=None, overwrite=False):
    if os.path.exists(fname) and not overwrite:
        raise OSError(f"File {fname} already exists. If you mean to replace it then use the argument \"overwrite=True\".")
    data = np.asarray(data, dtype=np.float64)
    cards = [_card("SIMPLE", True), _card("BITPIX", -64), _card("NAXIS", data.ndim)]
    for i, n in enumerate(reversed(data.shape)):
        cards.append(_card(f"NAXIS{i + 1}", n))
    cards.append(_card("EXTEND", True))
    for k, v in (header or {}).items():
        cards.append(_card(k, v))
    cards.append("END".ljust(80))
    hdr = "".join(cards).encode("ascii")
    hdr += b" " * (-len(hdr) % BLOCK)
    body = data.astype(">f8").tobytes()
    body += b"\0" * (-len(body) % BLOCK)
    with open(fname, "wb") as f:
        f.write(hdr + body)


def _parse_value(raw):
    raw = raw.split("/")[0].strip() if not raw.strip().startswith("'") else raw.strip()
    if raw.startswith("'"):
        end = raw.rfind("'")
        return raw[1:end].replace("''", "'").rstrip()
    if raw in ("T", "F"):
        return raw == "T"
    try:
        return int(raw)
    except ValueError:
        return float(raw)


def read(fname):
    """-> (header dict, float64 array)."""
    with open(fname, "rb") as f:
        buf = f.read()
    header, pos, done = {}, 0, False
    while not done:
        block = buf[pos:pos + BLOCK].decode("ascii")
        pos += BLOCK
        for i in range(0, BLOCK, 80):
            card = block[i:i + 80]
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] == "= ":
                header[key] = _parse_value(card[10:])
    shape = tuple(header[f"NAXIS{i}"] for i in range(header["NAXIS"], 0, -1))
    n = int(np.prod(shape)) if shape else 0
    data = np.frombuffer(buf, dtype=">f8", count=n, offset=pos).astype(np.float64).reshape(shape)
    return header, data


def getheader(fname):
    return read(fname)[0]


def getdata(fname):
    return read(fname)[1]
