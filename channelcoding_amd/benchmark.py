"""Command line of the reference's simulation driver (src/simulation/benchmark.c++) on the GPU path.

    python -m channelcoding_amd.benchmark [--simulation awgn|bitflip] [--algorithm NAME|all]...
                                          [--k 5|6|7|all]... [--dmin 3|5|7|9|11|all]... [--seed N | --seed-time]
                                          [--threads N] [--log-dir DIR]

Same options, same decoder registry (benchmark.c++:23-166: primitive_bch<k, dmin<d>, A> for k in 5..7,
d in 3,5,7,9 and the nine algorithm tags, min-sum family with 50 iterations, NMS 8/10, OMS 1/100) and the same
selection rule (intersection of the chosen names, powers and distances; an option that is never given selects
everything; an empty selection prints the usage and fails, :430-433).  Each selected decoder runs the chosen
simulation and writes "<to_string()>.log" in the reference's two-column format.

Where the reference spreads the decoders over a pool of CPU threads (--threads, :435-439), every simulation
here is batched on the GPU; launched under torch.distributed.run the frames of each Eb/N0 point are sharded over
the ranks' GPUs (montecarlo.awgn_simulation).  --threads is accepted and ignored.
"""
import argparse
import sys
import time

from . import codes as cc
from .montecarlo import awgn_simulation, bitflip_simulation

POWERS = (5, 6, 7)
DISTANCES = (3, 5, 7, 9)
ITERATIONS = 50
# to_string() suffix, lower-cased (benchmark.c++:212-216) -> tag factory
ALGORITHMS = {
    "bm": lambda: cc.berlekamp_massey_tag(),
    "pgz": lambda: cc.peterson_gorenstein_zierler_tag(),
    "euklid": lambda: cc.euklid_tag(),
    "ms": lambda: cc.min_sum_tag(ITERATIONS),
    "nms": lambda: cc.normalized_min_sum_tag(ITERATIONS, 8 / 10),
    "oms": lambda: cc.offset_min_sum_tag(ITERATIONS, 1 / 100),
    "scms1": lambda: cc.self_correcting_1_min_sum_tag(ITERATIONS),
    "scms2": lambda: cc.self_correcting_2_min_sum_tag(ITERATIONS),
    "2dnms": lambda: cc.normalized_2d_min_sum_tag(ITERATIONS),
}


_REGISTRY = None


def registry():
    """(name, k, template dmin, reported dmin) of every decoder the reference instantiates, in its order.
    The reference keys --dmin on the distance PRINTED by to_string() (benchmark.c++:224-230), which is the true
    BCH bound of the generator and can exceed the template argument: primitive_bch<5, dmin<9>> reports 11."""
    global _REGISTRY
    if _REGISTRY is None:
        from . import _capi as capi
        reg = []
        for name in ALGORITHMS:
            for k in POWERS:
                for d in DISTANCES:
                    text = cc.primitive_bch(k, cc.dmin(d), ALGORITHMS[name](), device=capi.DEVICE_NONE).to_string()
                    reg.append((name, k, d, int(text[text.rindex(" ") + 1:text.rindex(")")])))
        _REGISTRY = reg
    return _REGISTRY


def reported_distances():
    return sorted({r[3] for r in registry()})


def usage_text():
    lines = ["--simulation [awgn|bitflip]  Choose the simulation to run. The default is AWGN.",
             "--algorithm <name>           Choose algorithm:"]
    lines += ["  " + a for a in ALGORITHMS]
    lines += ["--k <num>                    Choose code length n = 2^k - 1;"] + ["  %d" % k for k in POWERS]
    lines += ["--dmin <num>                 Choose dmin of the code:"] + ["  %d" % d for d in reported_distances()]
    lines += ["--seed <num>                 Set seed of the random number generator. The default is 0.",
              "--seed-time                  Use the current time as seed for the random number generator.",
              "--threads <num>              Accepted for compatibility; simulations are batched on the GPU(s).",
              "--stop-rule <0|1|2>          Min-sum stop rule: 0 as shipped, 1 published, 2 GF(2) parity (default).",
              "--log-dir <dir>              Where the <decoder>.log files go (default: current directory).",
              "",
              "algorithm, k, and dmin can be specified multiple times.",
              "For all other options, giving them multiple times results in the last value being used."]
    return "\n".join(lines)


def select(algorithms, powers, distances):
    """benchmark.c++:374-428."""
    def chosen(values, reference, what, convert):
        out = set()
        for v in values or ():
            v = str(v).lower()
            if v == "all":
                return set(reference)
            try:
                key = convert(v)
            except ValueError:
                key = None
            if key not in reference:
                raise SystemExit("Unknown %s '%s'\n%s" % (what, v, usage_text()))
            out.add(key)
        return out or set(reference)

    names = chosen(algorithms, set(ALGORITHMS), "algorithm", str)
    ks = chosen(powers, set(POWERS), "k", int)
    ds = chosen(distances, set(reported_distances()), "dmin", int)
    return [(a, k, d) for a, k, d, shown in registry() if a in names and k in ks and shown in ds]


def build(name, k, d, stop_rule, device=None):
    kw = {} if device is None else {"device": device}
    return cc.primitive_bch(k, cc.dmin(d), ALGORITHMS[name](), stop_rule=stop_rule, **kw)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="channelcoding_amd.benchmark", add_help=False)
    ap.add_argument("--simulation", "-simulation", default="awgn")
    ap.add_argument("--algorithm", "-algorithm", action="append")
    ap.add_argument("--k", "-k", action="append")
    ap.add_argument("--dmin", "-dmin", action="append")
    ap.add_argument("--seed", "-seed", type=lambda v: int(v, 0), default=0)
    ap.add_argument("--seed-time", "-seed-time", action="store_true")
    ap.add_argument("--threads", "-threads", type=int, default=None)
    ap.add_argument("--stop-rule", type=int, default=2)
    ap.add_argument("--log-dir", default=".")
    ap.add_argument("--errors", type=int, default=0, help="bitflip: largest number of flipped bits")
    ap.add_argument("--max-samples", type=int, default=None, help="awgn: cap on frames per Eb/N0 point")
    ap.add_argument("--help", "-h", action="store_true")
    args, unknown = ap.parse_known_args(argv)
    if args.help or unknown:
        if unknown:
            print("Unkown argument: %s" % " ".join(unknown), file=sys.stderr)
        print(usage_text())
        return 1
    sim = args.simulation.lower()
    if sim not in ("awgn", "bitflip"):
        print("Don't know the simulation type '%s'" % sim, file=sys.stderr)
        print(usage_text())
        return 1
    chosen = select(args.algorithm, args.k, args.dmin)
    if not chosen:
        print("The selection is empty")
        print(usage_text())
        return 1
    seed = time.time_ns() if args.seed_time else args.seed

    import torch
    dist = None
    try:
        import os
        import torch.distributed as tdist
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            tdist.init_process_group("nccl", device_id=torch.device("cuda", local))
            dist = tdist
    except ImportError:
        pass
    rank = dist.get_rank() if dist else 0
    for name, k, d in chosen:
        code = build(name, k, d, args.stop_rule)
        t0 = time.perf_counter()
        if sim == "awgn":
            res = awgn_simulation(code, seed=seed, log_dir=args.log_dir, max_samples=args.max_samples)()
            frames = sum(r["frames"] for r in res)
        else:
            if rank == 0:
                res = bitflip_simulation(code, args.errors, log_dir=args.log_dir)()
                frames = sum(r["patterns"] for r in res)
            else:
                frames = 0
        if rank == 0:
            print("%s: %d frames in %.2f s" % (code.to_string(), frames, time.perf_counter() - t0), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
