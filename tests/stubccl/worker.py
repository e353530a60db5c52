"""One rank of the world-2 tests of the native collective path (tests/test_gpu_multirank.py): run as a child
process with TBNN_RCCL_LIB pointing at the stub collective library.  argv: mode rank world idfile outfile."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import tbnn_oracle as o                      # noqa: E402  (problem generator: this is a test)
from tensorbnn_amd import _native as nat    # noqa: E402
from tensorbnn_amd import parallel          # noqa: E402

SHAPES = {
    "narrow": ([5, 50, 50, 50, 1], 3000, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "wide": ([3, 20, 36, 2], 1500, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),
    "generic": ([4, 9, 2], 700, o.ACT_ELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "bern": ([20, 32, 16, 48, 2], 1030, o.ACT_SIGMOID, o.PRIOR_CAUCHY, o.LIK_BERNOULLI),
    "layered": ([40, 24, 24, 3], 900, o.ACT_TANH, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),       # three outputs: no fused family (kernels_layered.hpp)
}


def exchange_id(rank, idfile):
    if rank == 0:
        uid = nat.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
        return uid
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.01)
    return open(idfile, "rb").read()


def chain_of(spec, chain_id=0):
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    return nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=nat.KERNEL_AUTO, chain_id=chain_id)


def main():
    mode, rank, world, idfile, outfile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    res = {}
    if mode == "gather":
        # independent chains: chain_id = rank, different states; the gather must be chain-major
        spec, X, Y, theta, eta = o.synth_problem(*SHAPES["narrow"][:2])
        ch = chain_of(spec, chain_id=rank)
        ch.set_data(X, Y); ch.set_state(theta + np.float32(rank)); ch.set_hypers(eta * np.float32(1 + rank))
        comm = nat.Comm(ch, world, rank, exchange_id(rank, idfile))
        res["comm_count"] = np.array(comm.count())
        for it in range(3):                                  # a few transitions between gathers, as at checkpoint time
            ch.hmc_step(1e-5, 3)
            g = ch.gather_samples(comm)
            res[f"g{it}"] = g
            res[f"own{it}"] = np.concatenate([ch.get_state(), ch.get_hypers()])
        comm.close(); ch.close()
    elif mode == "shard":
        # one chain over `world` ranks: disjoint row blocks, identical theta / eta / seed
        uid = None
        for name, (dims, n, act, prior, lik) in SHAPES.items():
            spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
            ch = chain_of(spec)
            uid = exchange_id(rank, idfile + "." + name)
            comm = nat.Comm(ch, world, rank, uid)
            lo, hi = parallel.shard_rows(ch, X, Y, comm)
            ch.set_state(theta); ch.set_hypers(eta)
            import ctypes
            stub = ctypes.CDLL(os.environ["TBNN_RCCL_LIB"], mode=ctypes.RTLD_GLOBAL)      # the instance libtbnn resolved
            c0 = stub.stubccl_allreduce_calls()
            lp, g, st = ch.logp_grad(theta, eta)
            res[name + "_allreduce_per_pass"] = np.array(stub.stubccl_allreduce_calls() - c0)
            p0 = np.random.default_rng(5).standard_normal(spec.n_params).astype(np.float32)
            outs = [ch.hmc_step(1e-5, 4, p0=p0, log_u=-1e30, trace=True), ch.hmc_step(1e-5, 3)]      # injected, then free-running
            res[name + "_rows"] = np.array([lo, hi]); res[name + "_kernel"] = np.array(ch.kernel_name)
            res[name + "_lp"] = np.array(lp); res[name + "_g"] = g; res[name + "_st"] = np.array(st)
            res[name + "_trace"] = np.asarray(outs[0]["trace_logp"])
            res[name + "_lar"] = np.array([x["log_accept_ratio"] for x in outs]); res[name + "_acc"] = np.array([x["accepted"] for x in outs])
            res[name + "_theta"] = ch.get_state()
            if spec.n_hypers:                                # the hyper transition reads the all-reduced statistic
                h = ch.hyper_step(1e-5, 5, p0=np.random.default_rng(6).standard_normal(spec.n_hypers).astype(np.float32), log_u=-1e30)
                res[name + "_hlar"] = np.array(h["log_accept_ratio"]); res[name + "_eta"] = ch.get_hypers()
                # and the transition after it starts from the cached state refreshed for the new eta (no pass over the rows)
                o3 = ch.hmc_step(1e-5, 3, p0=p0, log_u=-1e30, trace=True)
                res[name + "_trace3"] = np.asarray(o3["trace_logp"]); res[name + "_theta3"] = ch.get_state()
            comm.close(); ch.close()
    else:
        raise SystemExit("unknown mode " + mode)
    np.savez(outfile, **res)


if __name__ == "__main__":
    main()
