"""The driver object: same ``network(...)``, ``add``, ``setupMCMC``, ``train``,
``predict`` API as the reference (tensorBNN/network.py:13-670).

What changes is the engine underneath: every epoch is one call to
``tbnn_hmc_step`` (the whole weight transition -- fused forward + likelihood +
gradient, leapfrog, Metropolis -- runs on the MI355X) plus, when
``adjustHypers`` is set, one ``tbnn_hyper_step``; the (eps, L) adapter is host
C++ (``tbnn_adapter_update``).  There is no CPU fallback.
"""
import os
import time

import numpy as np

from . import _native as nat
from .metrics import Accuracy, PercentError, SquaredError
from .paramAdapter import paramAdapter


class DualAveraging(object):
    """The hyper step-size adaptation of InnerStepHyper (network.py:457-469) with setupMCMC's constants
    (network.py:241-248): float32 arithmetic, target 0.95, gamma 0.4, t0 10, kappa 0.75, mu = log(100 eps0) and the
    exp(logEpsilonBar) applied while m < 0.8 burnin (Q6).  `network` runs one per chain; bench.py drives configs[4]'s
    hyper transition with the same object."""

    def __init__(self, hyperStepSize, burnin):
        f = np.float32
        self.target, self.gamma, self.t0, self.kappa = f(0.95), f(0.4), f(10), f(0.75)
        self.h, self.logEpsilonBar = f(0), f(0)
        self.mu = f(np.log(f(100 * hyperStepSize)))                          # :248
        self.burnin = burnin
        self.step_size = f(hyperStepSize)

    def update(self, epoch, log_accept_ratio):
        """epoch = iter_ before the increment; returns the acceptance probability it used (:459-460)"""
        f = np.float32
        m = f(epoch) + f(1)
        accept = f(np.exp(f(log_accept_ratio))) if log_accept_ratio < 0 else f(1)
        self.h = (f(1) - f(1) / (m + self.t0)) * self.h + (f(1) / (m + self.t0)) * (self.target - accept)
        logEpsilon = self.mu - self.h * (m ** f(0.5)) / self.gamma
        self.logEpsilonBar = (f(1) - m ** (-self.kappa)) * self.logEpsilonBar + m ** (-self.kappa) * logEpsilon
        if m < f(self.burnin * 0.8):
            self.step_size = f(np.exp(self.logEpsilonBar))
        return accept


class _SampleFiles(object):
    """The on-disk sample format of network.train (network.py:546-559, :609-663): <n>.<k>.txt per state tensor, hypers<k>.txt,
    architecture.txt, summary.txt, rotated every networksPerFile kept samples.  One instance per chain folder."""

    def __init__(self, filePath, states, layers):
        self.filePath = filePath
        os.makedirs(filePath, exist_ok=True)
        self.n_states = len(states)
        self.files = [open(filePath + "/" + str(n) + ".0" + ".txt", "wb") for n in range(self.n_states)]          # :549-552
        self.files.append(open(filePath + "/hypers" + "0" + ".txt", "wb"))
        with open(filePath + "/architecture.txt", "wb") as f:                                                     # :555-559
            for layer in layers:
                f.write((layer.name + "\n").encode("utf-8"))

    def after_epoch(self, iter_, startSampling, samplingStep, networksPerFile, states, hyperStates):
        """iter_: the epoch counter AFTER its increment (network.py:592)"""
        indexShift = iter_ - startSampling - 1                           # :609-646
        indexInterval = networksPerFile * samplingStep
        if iter_ > startSampling and indexShift % indexInterval == 0:
            for file in self.files:
                file.close()
            idx = int((iter_ - startSampling) // (networksPerFile * samplingStep))
            self.files = [open(self.filePath + "/" + str(n) + "." + str(idx) + ".txt", "wb") for n in range(self.n_states)]
            self.files.append(open(self.filePath + "/hypers" + str(idx) + ".txt", "wb"))
            with open(self.filePath + "/summary.txt", "wb") as file:
                for n in range(self.n_states):
                    file.write((" ".join(str(s) for s in states[n].shape).strip() + "\n").encode("utf-8"))
                numNetworks = indexShift // samplingStep
                numFiles = numNetworks // networksPerFile
                if numNetworks % networksPerFile != 0:
                    numFiles += 1
                file.write((str(numNetworks) + " " + str(numFiles) + " " + str(self.n_states) + "\n").encode("utf-8"))
                file.write(str(int(sum(h.size for h in hyperStates))).encode("utf-8"))
        if iter_ > startSampling and iter_ % samplingStep == 0:          # :648-663
            for n in range(len(self.files) - 1):
                np.savetxt(self.files[n], states[n])
            np.savetxt(self.files[-1], [np.reshape(h, (1,)) for h in hyperStates])

    def close(self):
        for file in self.files:
            file.close()


class network(object):
    def __init__(self, dtype, inputDims, trainX, trainY, validateX, validateY, device=0, chain_id=0, seed=50,
                 kernel=nat.KERNEL_AUTO):
        """network.py:19-58.  ``dtype`` must be float32 (the only type the reference's
        examples use and the type the kernels compute in).  Extra keyword arguments
        (device, chain_id, seed, kernel) select the GPU / the chain's Philox stream."""
        if np.dtype(dtype if not hasattr(dtype, "as_numpy_dtype") else dtype.as_numpy_dtype) != np.float32:
            raise TypeError("tensorbnn_amd computes in float32 only")
        self.dtype = np.float32
        self.inputDims = inputDims
        self.trainX = np.asarray(trainX, dtype=np.float32).reshape(len(trainX), inputDims)       # :41-44
        self.trainY = np.asarray(trainY, dtype=np.float32)                                        # :45
        self.validateX = np.asarray(validateX, dtype=np.float32).reshape(len(validateX), inputDims)
        self.validateY = np.asarray(validateY, dtype=np.float32)
        self.states = []          # :53  [W1, b1, W2, b2, ...] as [out,in] / [out,1] arrays
        self.hyperStates = []     # :54  list of shape-[1] arrays
        self.layers = []          # :56
        self.device, self.chain_id, self.seed, self.kernel = device, chain_id, seed, kernel
        self._chain = None
        self._dense = []          # descriptors (in, out, act, prior)
        self._lik_hypers = 0      # likelihood hyper entries at the tail of hyperStates (appended by train, once)

    # ------------------------------------------------------------------ model
    def add(self, layer, parameters=None):
        """network.py:173-191"""
        if layer.numTensors > 0:
            if getattr(layer, "prior_kind", None) is None:
                raise NotImplementedError(f"layer {layer.name!r} is not a dense layer the HIP path knows")
            self._dense.append([int(layer.inputDims), int(layer.outputDims), nat.ACT_NONE, int(layer.prior_kind)])
            for st in (layer.parameters if parameters is None else parameters):
                self.states.append(np.asarray(st, dtype=np.float32))
        else:
            if getattr(layer, "act_kind", None) is None:
                raise NotImplementedError(f"activation {layer.name!r} is not supported by the HIP path")
            if not self._dense or self._dense[-1][2] != nat.ACT_NONE or (self.layers and self.layers[-1].numTensors == 0):
                raise NotImplementedError("an activation must directly follow a dense layer (it is fused as its epilogue)")
            self._dense[-1][2] = int(layer.act_kind)
        self.layers.append(layer)
        if layer.numHyperTensors > 0:
            # a layer added after a first train(): the likelihood's hypers sit at the tail of hyperStates (appended by
            # train); take them off so that the layer's rows go where the eta layout expects them -- train() re-appends
            if self._lik_hypers:
                del self.hyperStates[-self._lik_hypers:]
                self._lik_hypers = 0
            for row in np.asarray(layer.hypers, dtype=np.float32):          # :189-191: one [1]-tensor per row
                self.hyperStates.append(np.asarray(row, dtype=np.float32).reshape(1))
        if self._chain is not None:       # the architecture changed: the old chain's device buffers go now, not at GC time
            self._chain.close()
        self._chain = None

    def _theta(self):
        return np.concatenate([np.asarray(s, dtype=np.float32).reshape(-1) for s in self.states])

    def _set_states_from(self, theta):
        o = 0
        for i, s in enumerate(self.states):
            self.states[i] = theta[o:o + s.size].reshape(s.shape).copy()
            o += s.size

    def _ensure_chain(self, likelihood=None):
        lik = likelihood if likelihood is not None else getattr(self, "likelihood", None)
        kind = lik.kind if lik is not None else nat.LIK_FIXED_GAUSSIAN
        sd = float(getattr(lik, "fixed_sd", 0.1)) if lik is not None else 0.1
        key = (tuple(map(tuple, self._dense)), kind, sd)
        if self._chain is None or self._chain_key != key:
            if self._chain is not None:
                self._chain.close()
            self._chain = nat.Chain(self._dense, likelihood=kind, fixed_sd=sd, device=self.device, seed=self.seed,
                                    chain_id=self.chain_id, kernel=self.kernel)
            self._chain_key = key
            y = self.trainY.reshape(len(self.trainX), -1)
            self._chain.set_data(self.trainX, y)
            if self.validateX is not None and len(self.validateX):          # network.py:47-51: staged once
                self._chain.set_validation(self.validateX, np.asarray(self.validateY).reshape(len(self.validateX), -1))
        return self._chain

    def predict(self, train, *argv):
        """network.py:141-171: [d_out, n] prediction over the rows staged on the device (tbnn_predict)."""
        tensors = self.states if len(argv) == 0 else argv[0]
        theta = np.concatenate([np.asarray(s, dtype=np.float32).reshape(-1) for s in tensors])
        ch = self._ensure_chain()
        if not train and not getattr(ch, "nv", 0):
            return ch.forward(self.validateX, theta)
        return ch.predict(0 if train else 1, theta)

    def _metrics_on_device(self):
        """The three reference metrics straight from the device (tbnn_metrics: forward + reduction, three doubles
        back); False when the list holds anything else or no validation rows are staged -- then the predictions
        go through `metrics` as in the reference."""
        ch = self._ensure_chain()
        if not getattr(ch, "nv", 0) or not all(type(m) in (SquaredError, PercentError, Accuracy) for m in self.metricList):
            return False
        theta = self._theta()
        for m in self.metricList:
            # metrics.py:44-47: SquaredError does not exponentiate the validation predictions
            pv_exp = bool(m.scaleExp) and not isinstance(m, SquaredError)
            tr = ch.metrics(0, theta, m.mean, m.sd, bool(m.scaleExp), bool(m.scaleExp))
            va = ch.metrics(1, theta, m.mean, m.sd, pv_exp, bool(m.scaleExp))
            if isinstance(m, SquaredError):
                m.squaredErrorTrain, m.squaredErrorValidate = tr[0], va[0]
            elif isinstance(m, PercentError):
                m.percentErrorTrain, m.percentErrorValidate = tr[1], va[1]
            else:
                m.accuracyTrain, m.accuracyValidate = 1.0 - tr[2], 1.0 - va[2]
            m.display()
        return True

    def metrics(self, trainPredict, trainReal, validatePredict, validateReal):
        """network.py:60-82"""
        for metric in self.metricList:
            metric.calculate(trainPredict, validatePredict, trainReal, validateReal)
            metric.display()

    # ------------------------------------------------------------------ MCMC
    def setupMCMC(self, stepSizeStart=1e-3, stepSizeMin=1e-4, stepSizeMax=1e-2, stepSizeOptions=40,
                  leapfrogStart=1000, leapfogMin=100, leapFrogMax=10000, leapfrogIncrement=1, hyperStepSize=1e-2,
                  hyperLeapfrog=100, burnin=1000, cores=4, averagingSteps=10, a=4, delta=0.1, strikes=5,
                  randomSteps=10, dualAveraging=False, adapt=True):
        """network.py:193-278 (argument names, including the misspelt ones, are the API).
        ``adapt=False`` (new) bypasses the (eps, L) adapter: fixed stepSizeStart / leapfrogStart."""
        # (kept: trainChains builds one adapter per chain from the same arguments)
        self._adapt_args = (stepSizeStart, leapfrogStart, stepSizeMin, stepSizeMax, stepSizeOptions, leapfogMin, leapFrogMax,
                            leapfrogIncrement, averagingSteps, burnin / averagingSteps)
        self._adapt_kw = dict(a=a, delta=delta, cores=cores, strikes=strikes, randomSteps=randomSteps)
        self._hyperStepSize0 = hyperStepSize
        self.adapt = self._make_adapter(self.chain_id)
        self.adapt_enabled = adapt
        self.step_size = np.float32(stepSizeStart)
        self.leapfrog = np.int32(leapfrogStart)
        self.cores = cores
        self.burnin = burnin
        self._da = DualAveraging(hyperStepSize, burnin)   # :241-248
        self.dualAveraging = dualAveraging
        self.hyperLeapfrog = hyperLeapfrog

    def _make_adapter(self, chain_id):
        """the (eps, L) adapter of the chain with this id (network.py:221-235); its own random stream per chain"""
        return paramAdapter(*self._adapt_args, seed=self.seed + 7919 * chain_id, **self._adapt_kw)

    # the reference keeps these scalars on the network object (network.py:241-248)
    hyper_step_size = property(lambda self: self._da.step_size)
    h = property(lambda self: self._da.h)
    logEpsilonBar = property(lambda self: self._da.logEpsilonBar)
    target = property(lambda self: self._da.target)
    mu = property(lambda self: self._da.mu)

    def _dual_averaging(self, epoch, log_accept_ratio):
        """network.py:457-469 (epoch = iter_ before the increment)."""
        return self._da.update(epoch, log_accept_ratio)

    def train(self, epochs, samplingStep, likelihood, metricList=[], adjustHypers=True, scaleExp=False,
              folderName=None, networksPerFile=1000, displaySkip=1, verbose=True, gather=None):
        """network.py:509-670.  Returns a list of per-epoch records (new; the reference
        only prints).  ``gather``: optional callable(theta_eta_device_ptr) used by
        tensorbnn_amd.parallel for the RCCL sample gather at checkpoint time."""
        startSampling = self.burnin
        self.likelihood = likelihood
        self.makeResponseLikelihood = likelihood.makeResponseLikelihood
        self.metricList = metricList
        self.adjustHypers = adjustHypers
        # :542-543 appends the likelihood's hypers on EVERY call, so a second train() (a continued run) hands H+1 values
        # to the sampler; here they are appended once and replaced when the likelihood changes
        if self._lik_hypers:
            del self.hyperStates[-self._lik_hypers:]
        for val in likelihood.hypers:
            self.hyperStates.append(np.asarray(val, dtype=np.float32).reshape(1))
        self._lik_hypers = len(likelihood.hypers)
        ch = self._ensure_chain(likelihood)
        if verbose:
            print("tensorbnn_amd: fused kernel", ch.kernel_name)
        ch.set_state(self._theta())
        ch.set_hypers(np.concatenate(self.hyperStates) if self.hyperStates else np.zeros(0, np.float32))

        writer = None
        if folderName is not None:                                           # :546-559
            writer = _SampleFiles(os.path.join(os.getcwd(), folderName), self.states, self.layers)

        iter_ = 0
        self.mainAccept = np.float32(0)
        self.hyperAccept = np.float32(0)
        records = []
        startTime = time.time()
        while iter_ < epochs:                                                # :567
            out = ch.hmc_step(float(self.step_size), int(self.leapfrog))     # InnerStepMain :368-412
            self.mainAccept = np.float32(out["accept_prob"])
            rec = {"iter": iter_, "eps": float(self.step_size), "L": int(self.leapfrog), "main": out}
            if self.adjustHypers and ch.H > 0:                               # InnerStepHyper :414-471
                hout = ch.hyper_step(float(self.hyper_step_size), int(self.hyperLeapfrog))
                self.hyperAccept = self._dual_averaging(iter_, hout["log_accept_ratio"])
                rec["hyper"] = hout
                rec["hyper_step_size"] = float(self.hyper_step_size)
            theta = ch.get_state()
            self._set_states_from(theta)
            eta = ch.get_hypers()
            self.hyperStates = [eta[i:i + 1].copy() for i in range(eta.size)]
            iter_ += 1
            if verbose and iter_ % displaySkip == 0:                         # :593-602
                print()
                print("iter:{:>2}".format(iter_))
                print("step size", self.step_size)
                print("hyper step size", self.hyper_step_size)
                print("leapfrog", self.leapfrog)
                print("Main acceptance", self.mainAccept)
                print("Hyper acceptance", self.hyperAccept)
                if not self._metrics_on_device():
                    self.metrics(self.predict(True), self.trainY, self.predict(False), self.validateY)
            if self.adapt_enabled:                                           # :603-607
                step, leap = self.adapt.update(self.states)
                self.step_size, self.leapfrog = np.float32(step), np.int32(leap)
                rec["sjd"] = self.adapt.lastSJD

            if writer is not None:                                           # :609-663
                writer.after_epoch(iter_, startSampling, samplingStep, networksPerFile, self.states, self.hyperStates)
            if gather is not None and iter_ > startSampling and iter_ % samplingStep == 0:
                gather(ch, iter_)
            if verbose and iter_ % displaySkip == 0:                         # :664-667
                likelihood.display(self.hyperStates)
                print("Time elapsed:", time.time() - startTime)
                startTime = time.time()
            records.append(rec)
        if writer is not None:
            writer.close()
        return records

    def trainChains(self, chains, epochs, samplingStep, likelihood, adjustHypers=True, folderName=None, networksPerFile=1000,
                    verbose=False, initialStates=None):
        """NEW (no counterpart in the reference, which runs one chain per process): `chains` independent chains of this network
        on the one GPU behind one native handle (tbnn_create_multi).  Chain c IS the run `network(..., chain_id=chain_id + c)`
        + `train(...)` would be: its own Philox stream, its OWN (eps, L) adapter (paramAdapter, network.py:221-235, :603-607)
        and its OWN dual averaging of the hyper step size (network.py:457-469) -- bit for bit, states and records
        (tests/test_gpu_multichain.py).  The chains advance in lockstep launches (gridDim.y = chain) for max_c L_c leapfrog
        steps per epoch; a chain past its own L_c is skipped on the device (tbnn_hmc_step_each).
        initialStates: None (every chain starts from the network's current state, as C runs of one script would) or an array
        [chains][P] of flattened start states -- over-dispersed starts are what R-hat style diagnostics want.
        Samples go to folderName/chain<c>/ in the reference's format (readable by `predictor`).  For problems that leave the
        GPU idle -- the reference's examples: tens to thousands of rows -- C chains cost about what one costs.  Returns the
        per-epoch records: rec["main"] / rec["hyper"]: one transition record per chain, rec["eps"] / rec["L"] /
        rec["hyper_step_size"]: per-chain lists.  The network object mirrors chain 0 (states, step size, leapfrog)."""
        startSampling = self.burnin
        self.likelihood = likelihood
        if self._lik_hypers:
            del self.hyperStates[-self._lik_hypers:]
        for val in likelihood.hypers:
            self.hyperStates.append(np.asarray(val, dtype=np.float32).reshape(1))
        self._lik_hypers = len(likelihood.hypers)
        grp = nat.ChainGroup(self._dense, int(chains), likelihood=likelihood.kind, fixed_sd=float(getattr(likelihood, "fixed_sd", 0.1)),
                             device=self.device, seed=self.seed, chain_id=self.chain_id, kernel=self.kernel)
        C = grp.C
        if verbose:
            print("tensorbnn_amd: fused kernel", grp.kernel_name, "x", C, "chains")
        grp.set_data(self.trainX, self.trainY.reshape(len(self.trainX), -1))
        if initialStates is None:
            grp.set_state(self._theta())
        else:
            th0 = np.asarray(initialStates, dtype=np.float32)
            if th0.shape != (C, grp.P):
                grp.close()
                raise ValueError(f"initialStates must be [{C}, {grp.P}]")
            grp.set_state(th0)
        grp.set_hypers(np.concatenate(self.hyperStates) if self.hyperStates else np.zeros(0, np.float32))
        shapes = [s_.shape for s_ in self.states]

        def split(theta):
            out, o = [], 0
            for shp in shapes:
                size = int(np.prod(shp))
                out.append(theta[o:o + size].reshape(shp).copy())
                o += size
            return out

        # one adapter and one dual averaging per chain; chain 0's are the network's own (what train() would use)
        adapters = [self.adapt] + [self._make_adapter(self.chain_id + c) for c in range(1, C)]
        das = [self._da] + [DualAveraging(self._hyperStepSize0, self.burnin) for _ in range(1, C)]
        eps = np.full(C, self.step_size, dtype=np.float32)
        leap = np.full(C, self.leapfrog, dtype=np.int32)
        writers = None
        if folderName is not None:
            writers = [_SampleFiles(os.path.join(os.getcwd(), folderName, "chain%d" % c), self.states, self.layers) for c in range(C)]
        records, iter_ = [], 0
        try:
            while iter_ < epochs:
                outs = grp.hmc_step_each(eps, leap)
                rec = {"iter": iter_, "eps": [float(e) for e in eps], "L": [int(l) for l in leap], "main": outs}
                if adjustHypers and grp.H > 0:
                    eps_h = np.array([float(d.step_size) for d in das], dtype=np.float32)
                    houts = grp.hyper_step_each(eps_h, int(self.hyperLeapfrog))
                    for c in range(C):
                        acc = das[c].update(iter_, houts[c]["log_accept_ratio"])
                        if c == 0:
                            self.hyperAccept = acc
                    rec["hyper"] = houts
                    rec["hyper_step_size"] = [float(d.step_size) for d in das]
                thetas, etas = grp.get_state(), grp.get_hypers()
                self._set_states_from(thetas[0])                          # the network object mirrors chain 0
                self.hyperStates = [etas[0][i:i + 1].copy() for i in range(etas.shape[1])]
                self.mainAccept = np.float32(outs[0]["accept_prob"])
                iter_ += 1
                if self.adapt_enabled:
                    sjd = []
                    for c in range(C):
                        e_c, l_c = adapters[c].update(thetas[c])
                        eps[c], leap[c] = np.float32(e_c), np.int32(l_c)
                        sjd.append(adapters[c].lastSJD)
                    rec["sjd"] = sjd
                    self.step_size, self.leapfrog = np.float32(eps[0]), np.int32(leap[0])
                if writers is not None:
                    for c, w in enumerate(writers):
                        w.after_epoch(iter_, startSampling, samplingStep, networksPerFile, split(thetas[c]),
                                      [etas[c][i:i + 1] for i in range(etas.shape[1])])
                records.append(rec)
        finally:
            if writers is not None:
                for w in writers:
                    w.close()
            grp.close()
        return records
