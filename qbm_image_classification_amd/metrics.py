"""Minimal counterpart of the reference's src/metrics.py for the Disc_QBM mirror: the History container
(src/metrics.py:11-21), get_metrics (:190-202) and save_history (:239-259).  Plotting and the result-text
writer are reporting, out of scope."""
from __future__ import annotations

import pickle


class History:
    def __init__(self, loss_per_batch, loss_per_epoch, nll_per_batch, nll_per_epoch, acc_per_epoch, auc_per_epoch,
                 combined_acc_auc_per_epoch):
        self.errors_per_batch = loss_per_batch
        self.error_per_epoch = loss_per_epoch
        self.nll_per_batch = nll_per_batch
        self.nll_per_epoch = nll_per_epoch
        self.distribution_per_epoch = []
        self.acc_per_epoch = acc_per_epoch
        self.auc_per_epoch = auc_per_epoch
        self.combined_acc_auc_per_epoch = combined_acc_auc_per_epoch


def get_metrics(y_true, y_predict, class_titles):
    """-> (accuracy, f1, precision, recall, auc), the tuple order Disc_QBM.train_model unpacks
    (reference contract: src/metrics.py:190-202).  Two classes: positive-class scores and the ROC AUC of the HARD
    predictions; more classes: macro averages and auc = 0."""
    from sklearn import metrics as skm

    binary = len(class_titles) == 2
    avg = "binary" if binary else "macro"
    scores = {fn.__name__: fn(y_true, y_predict, average=avg, **kw)
              for fn, kw in ((skm.f1_score, {}), (skm.precision_score, {"zero_division": 0}), (skm.recall_score, {}))}
    auc = skm.roc_auc_score(y_true, y_predict) if binary else 0
    return (skm.accuracy_score(y_true, y_predict), scores["f1_score"], scores["precision_score"],
            scores["recall_score"], auc)


def save_history(file_path_and_name, history: History):
    for name, values in (("loss_per_batch", history.errors_per_batch), ("loss_per_epoch", history.error_per_epoch),
                         ("nll_per_batch", history.nll_per_batch), ("nll_per_epoch", history.nll_per_epoch),
                         ("acc_per_epoch", history.acc_per_epoch), ("auc_per_epoch", history.auc_per_epoch),
                         ("combined_acc_auc_per_epoch", history.combined_acc_auc_per_epoch)):
        with open(file_path_and_name + name + ".pkl", "wb") as f:
            pickle.dump(values, f)
