"""Metrics printed during training (reference: tensorBNN/metrics.py); host-side
reporting on the predictions the native forward kernel returns."""
import numpy as np


class Metric(object):
    def __init__(self, scaleExp=False, mean=0, sd=1, *argv, **kwargs):
        self.scaleExp = scaleExp
        self.mean = mean
        self.sd = sd

    def calculate(self, predictionsTrain, predictionsValidate, realTrain, realValidate, *argv, **kwargs):
        pass

    def display(self):
        pass

    def _prep(self, predictionsTrain, predictionsValidate, realTrain, realValidate, exp_train_pred=True):
        pt = np.asarray(predictionsTrain).T * self.sd + self.mean           # metrics.py:36-39
        pv = np.asarray(predictionsValidate).T * self.sd + self.mean
        rt = np.asarray(realTrain) * self.sd + self.mean
        rv = np.asarray(realValidate) * self.sd + self.mean
        if self.scaleExp:
            pt, rt, rv = np.exp(pt), np.exp(rt), np.exp(rv)
            if exp_train_pred:
                pv = np.exp(pv)
        return pt, pv, rt.reshape(pt.shape), rv.reshape(pv.shape)


class SquaredError(Metric):
    """metrics.py:30-67 (note: with scaleExp the reference does not exponentiate
    the validation predictions, :44-47 -- kept)."""

    def calculate(self, predictionsTrain, predictionsValidate, realTrain, realValidate):
        pt, pv, rt, rv = self._prep(predictionsTrain, predictionsValidate, realTrain, realValidate, exp_train_pred=False)
        self.squaredErrorTrain = float(np.mean((pt - rt) ** 2))
        self.squaredErrorValidate = float(np.mean((pv - rv) ** 2))

    def display(self):
        print("training squared error{: 9.5f}".format(self.squaredErrorTrain),
              "validation squared error{: 9.5f}".format(self.squaredErrorValidate))


class PercentError(Metric):
    """metrics.py:70-108"""

    def calculate(self, predictionsTrain, predictionsValidate, realTrain, realValidate):
        pt, pv, rt, rv = self._prep(predictionsTrain, predictionsValidate, realTrain, realValidate)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.percentErrorTrain = float(np.mean(np.abs((pt - rt) / rt) * 100))
            self.percentErrorValidate = float(np.mean(np.abs((pv - rv) / rv) * 100))

    def display(self):
        print("training percent error{: 7.3f}".format(self.percentErrorTrain),
              "validation percent error{: 7.3f}".format(self.percentErrorValidate))


class Accuracy(Metric):
    """metrics.py:110-141"""

    def calculate(self, predictionsTrain, predictionsValidate, realTrain, realValidate):
        pt, pv, rt, rv = self._prep(predictionsTrain, predictionsValidate, realTrain, realValidate)
        self.accuracyTrain = float(1 - np.mean(np.abs(rt - np.round(pt))))
        self.accuracyValidate = float(1 - np.mean(np.abs(rv - np.round(pv))))

    def display(self):
        print("training accuracy{: 9.5f}".format(self.accuracyTrain),
              "validation accuracy{: 9.5f}".format(self.accuracyValidate))
