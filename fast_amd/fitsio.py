"""Minimal FITS primary-HDU writer / reader (float64 image + scalar header cards).

The reference saves results with astropy (`fits.writeto`, fast/fast.py:809-812) and reloads them
with `fits.getheader` / `fits.getdata` (fast.py:998-1002); astropy is not installed here, so this
module writes the same standard file by hand: 80-character cards, 2880-byte blocks, big-endian
BITPIX = -64 data.  Files written here open in astropy and vice versa (single primary HDU)."""
import os

import numpy as np

BLOCK = 2880


def _card(key, value):
    key = key.upper()[:8]
    if isinstance(value, bool):
        v = f"{'T' if value else 'F':>20}"
    elif isinstance(value, (int, np.integer)):
        v = f"{int(value):>20d}"
    elif isinstance(value, (float, np.floating)):
        v = f"{repr(float(value)).upper():>20}" if np.isfinite(value) else f"'{value}'"
    else:
        s = str(value).replace("'", "''")
        v = f"'{s:<8}'"
    return f"{key:<8}= {v}".ljust(80)[:80]


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
