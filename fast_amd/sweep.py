"""Many-configuration sweeps (the pattern of fast/complete_orbit_simulation.py:187-232 of the
reference: one `Fast` object per geometry sample, few iterations each).

BASELINE config 5 is a zenith-angle scan: 32 Cn2 geometries x 4096 iterations at N = 1024, AO
corrected.  Per sample the reference rebuilds everything (`fast.Fast(params)`,
complete_orbit_simulation.py:227) and its init is dominated by `compute_powerspec` (12-15 s at
1024^2).  Here the power spectrum is one GPU kernel (< 1 ms), the pupil /
fibre-mode products are cached across samples that share the aperture (host.pupils cache), and
samples are dealt round-robin to the ranks of a multi-process launch (one process per GPU:
`zenith_scan(..., rank, world)`) and, inside a process that drives several GPUs, to one thread per device
(`devices=[...]`); only the per-sample summary statistics are gathered, as JSON --
nothing received from another rank is ever unpickled.
"""
import base64
import copy
import json
import time

import numpy


from .fast import Fast


def zenith_scan(base_params, zenith_angles, niter=4096, nchunks=1, keep_power=False, rank=0, world=1, devices=None):
    """Run `Fast` for each zenith angle owned by this rank (angles[rank::world]).

    `devices`: the GPUs this process drives.  With several (one process, N devices: `bench.py --gpus N` without a launcher,
    `GPU_DEVICES` of the caller) the rank's samples are dealt round-robin to one thread per device -- BASELINE configs[4]
    is "4 configs per GPU" on 8 GPUs in either launch form; ctypes releases the GIL inside the library, every handle has
    its own stream and the library's per-device state is locked.  None / one entry: `GPU_DEVICE` of base_params (or that entry).

    L_SAT is recomputed from H_SAT for each angle (funcs.l_path via Fast.init_atmos,
    fast.py:237-241) because base_params['L_SAT'] is left None.  Returns a list of dicts
    (zenith, mean dB_rel, scintillation index, phs_var, logamp_var, timings, device), ordered by sample index."""
    mine = list(range(rank, len(zenith_angles), world))
    devs = None if devices is None else [int(d) for d in devices]

    def one(idx, device):
        p = copy.copy(base_params)
        p.update({"ZENITH_ANGLE": float(zenith_angles[idx]), "NITER": niter, "NCHUNKS": nchunks, "TEMPORAL": False,
                  "GPU_SHARD": False})
        if device is not None:
            p["GPU_DEVICE"] = device
            p["GPU_DEVICES"] = None
        if p.get("SEED") is not None:
            p["SEED"] = int(p["SEED"]) + idx
        t0 = time.perf_counter()
        sim = Fast(p)
        t1 = time.perf_counter()
        res = sim.run()
        t2 = time.perf_counter()
        rec = {"index": idx, "zenith": float(zenith_angles[idx]), "mean_dB_rel": float(res.avg_power_dB_rel),
               "scintillation_index": float(res.scintillation_index), "phs_var": float(sim.phs_var),
               "logamp_var": float(sim.logamp_var), "r0_los": float(sim.r0_los), "L": float(sim.L),
               "init_s": t1 - t0, "run_s": t2 - t1, "powerspec_kernel_ms": sim.powerspec_kernel_ms, "device": int(sim.device)}
        if keep_power:
            rec["r"] = res._r
        return rec

    if not devs or len(devs) == 1:
        return [one(idx, devs[0] if devs else None) for idx in mine]
    import threading
    out, errs = [], []

    def worker(k):
        try:
            for idx in mine[k::len(devs)]:
                out.append(one(idx, devs[k]))          # (list.append is atomic under the GIL)
        except BaseException as e:                     # noqa: BLE001 -- re-raised in the caller's thread
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(k,), name=f"fastmc-sweep-{k}") for k in range(len(devs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return sorted(out, key=lambda r: r["index"])


def _encode(records):
    """Records -> JSON bytes; numpy arrays as {dtype, shape, base64 of the raw bytes} (plain data, no pickle)."""
    def enc(v):
        if isinstance(v, numpy.ndarray):
            a = numpy.ascontiguousarray(v)
            if a.dtype.hasobject:
                raise TypeError("object arrays cannot be exchanged")
            return {"__ndarray__": a.dtype.str, "shape": list(a.shape), "data": base64.b64encode(a.tobytes()).decode("ascii")}
        if isinstance(v, numpy.generic):
            return v.item()
        return v
    return json.dumps([{k: enc(v) for k, v in r.items()} for r in records]).encode()


def _decode(blob):
    def dec(v):
        if isinstance(v, dict) and "__ndarray__" in v:
            dt = numpy.dtype(v["__ndarray__"])
            if dt.hasobject:
                raise ValueError("object arrays are not accepted")
            return numpy.frombuffer(base64.b64decode(v["data"]), dtype=dt).reshape(v["shape"]).copy()
        return v
    recs = json.loads(blob.decode())
    if not isinstance(recs, list) or not all(isinstance(r, dict) for r in recs):
        raise ValueError("malformed sweep records")
    return [{k: dec(v) for k, v in r.items()} for r in recs]


def gather_records(records):
    """All ranks' records on every rank (through the rendezvous of a multi-rank launch, fast_amd/rendezvous.py)."""
    from . import rendezvous
    rdzv = rendezvous.from_env()
    if rdzv is None:
        return sorted(records, key=lambda r: r["index"])
    parts = rdzv.exchange(_encode(records))
    return sorted([r for part in parts for r in _decode(part)], key=lambda r: r["index"])
