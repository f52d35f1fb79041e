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
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score, roc_auc_score

    acc = accuracy_score(y_true, y_predict)
    if len(class_titles) == 2:
        f1 = f1_score(y_true, y_predict, average='binary')
        precision = precision_score(y_true, y_predict, average='binary', zero_division=0)
        recall = recall_score(y_true, y_predict, average='binary')
        auc = roc_auc_score(y_true, y_predict)   # on hard predictions, like the reference
    else:
        f1 = f1_score(y_true, y_predict, average='macro')
        precision = precision_score(y_true, y_predict, average='macro', zero_division=0)
        recall = recall_score(y_true, y_predict, average='macro')
        auc = 0
    return acc, f1, precision, recall, auc


def save_history(file_path_and_name, history: History):
    for name, values in (("loss_per_batch", history.errors_per_batch), ("loss_per_epoch", history.error_per_epoch),
                         ("nll_per_batch", history.nll_per_batch), ("nll_per_epoch", history.nll_per_epoch),
                         ("acc_per_epoch", history.acc_per_epoch), ("auc_per_epoch", history.auc_per_epoch),
                         ("combined_acc_auc_per_epoch", history.combined_acc_auc_per_epoch)):
        with open(file_path_and_name + name + ".pkl", "wb") as f:
            pickle.dump(values, f)
