"""Checkpoint I/O and forward-only feature extraction on the MI355X kernels.

Same public names and behaviour as /root/reference/cpc/feature_loader.py -- FeatureModule (:15-54), CPCModule (:57-82),
getCheckpointData (:176-199), loadModel (:238-283, single-checkpoint case), get_module (:286-293), save_checkpoint
(:296-304), seqNormalization (:316-320), buildFeature (:323-367), buildFeature_batch (:370-433) -- built differently:
feature extraction is a PLAN (which sample ranges of the file go through the model, and how many trailing frames of each
are kept) executed by one driver that keeps the waveform on the device and feeds equal-length ranges to the encoder as one
batch where the reference's semantics allow it.  Checkpoints use the reference's layout {"gEncoder", "cpcCriterion",
"optimizer", "best"} and key names, so files written by either side load in the other.
"""
import argparse
import json
import os
from collections import namedtuple

import torch

from . import audio
from .model import CPCModel
from .train import getAR, getEncoder


# --------------------------------------------------------------------------- model wrappers
def _model_device(module):
    return next(module.parameters()).device


class FeatureModule(torch.nn.Module):
    """feature_loader.py:15-54: `forward((audio, label))` -> context features (or encoder outputs with get_encoded),
    optionally collapsed to [frames, dim].  The sklearn CCA projection of the reference is not on this path."""

    def __init__(self, featureMaker, get_encoded, collapse=False, cca_projection=None):
        super().__init__()
        if cca_projection:
            raise NotImplementedError("cca_projection is not on the MI355X path")
        self.featureMaker = featureMaker
        self.get_encoded = get_encoded
        self.collapse = collapse
        self.cca_projection = None

    @property
    def out_feature_dim(self):
        net = self.featureMaker.gEncoder if self.get_encoded else self.featureMaker.gAR
        return net.getDimOutput()

    def getDownsamplingFactor(self):
        return self.featureMaker.gEncoder.DOWNSAMPLING

    def forward(self, data):
        waves, label = data
        if waves.dim() == 4:                                # [b, 2, 1, samples] training batches: the first view
            waves = waves[:, 0]
        context, encoded, _ = self.featureMaker(waves.to(_model_device(self.featureMaker)), label)
        feats = encoded if self.get_encoded else context
        return feats.contiguous().view(-1, feats.size(2)) if self.collapse else feats


class CPCModule(torch.nn.Module):
    """feature_loader.py:57-82: the criterion's view of an utterance -- softmax over the 1 + negatives candidates of
    prediction step n_pred ([b, 1 + negatives, W]), or, with main_distance_only, the positive's score alone ([b, 1, W])."""

    def __init__(self, feature_maker, cpc_criterion, main_distance_only=False, n_pred=-1):
        super().__init__()
        self.feature_maker = feature_maker
        self.cpc_criterion = cpc_criterion
        self.main_distance_only = main_distance_only
        self.n_pred = n_pred

    def getDownsamplingFactor(self):
        return self.feature_maker.gEncoder.DOWNSAMPLING

    def forward(self, data):
        waves, label = data
        context, encoded, label = self.feature_maker(waves.to(_model_device(self.feature_maker)), label)
        if self.main_distance_only:
            return self.cpc_criterion.getCosineDistances(context, encoded)[self.n_pred]
        scores, _ = self.cpc_criterion.getPrediction(context, encoded, label)
        return torch.softmax(scores[self.n_pred], dim=1)


def get_module(i_module):
    """The CPCModel under any stack of DataParallel / DistributedDataParallel / FeatureModule wrappers (:286-293)."""
    while True:
        if isinstance(i_module, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
            i_module = i_module.module
        elif isinstance(i_module, FeatureModule):
            i_module = i_module.featureMaker
        else:
            return i_module


# --------------------------------------------------------------------------- checkpoints
def save_checkpoint(model_state, criterion_state, optimizer_state, best_state, path_checkpoint):
    torch.save(dict(gEncoder=model_state, cpcCriterion=criterion_state, optimizer=optimizer_state, best=best_state),
               path_checkpoint)


def _checkpoint_index(name):
    """N of 'checkpoint_N.pt', else None."""
    stem, ext = os.path.splitext(name)
    return int(stem[11:]) if ext == ".pt" and stem[11:].isdigit() else None


def getCheckpointData(pathDir):
    """(path of the newest checkpoint_N.pt, logs, args namespace) of a run directory, None without one (:176-199)."""
    if not os.path.isdir(pathDir):
        return None
    numbered = [(idx, name) for name in os.listdir(pathDir) for idx in [_checkpoint_index(name)] if idx is not None]
    if not numbered:
        return None
    newest = max(numbered)[1]
    with open(os.path.join(pathDir, "checkpoint_logs.json")) as f:
        logs = json.load(f)
    with open(os.path.join(pathDir, "checkpoint_args.json")) as f:
        args = argparse.Namespace(**json.load(f))
    return os.path.abspath(os.path.join(pathDir, newest)), logs, args


def loadModel(pathCheckpoints, loadStateDict=True):
    """CPCModel built from the run's checkpoint_args.json, `gEncoder` weights loaded (:238-283; one checkpoint, no
    nested `load` chains, no ConcatenatedModel).  Returns (model, hiddenGar, hiddenEncoder)."""
    if len(pathCheckpoints) != 1:
        raise NotImplementedError("ConcatenatedModel (several checkpoints) is not on the MI355X path")
    (path,) = pathCheckpoints
    _, _, run_args = getCheckpointData(os.path.dirname(path))
    model = CPCModel(getEncoder(run_args), getAR(run_args))
    if loadStateDict:
        model.load_state_dict(torch.load(path, "cpu")["gEncoder"], strict=False)
    return model, run_args.hiddenGar, run_args.hiddenEncoder


# --------------------------------------------------------------------------- feature extraction
def seqNormalization(out):
    """[batch, frames, channels] -> zero mean, unit (unbiased) variance along the frames (:316-320)."""
    centred = out - out.mean(dim=1, keepdim=True)
    return centred / torch.sqrt(out.var(dim=1, keepdim=True) + 1e-08)


# a range of samples that goes through the model; `tail` > 0 keeps only its last `tail` frames
_Span = namedtuple("_Span", "start stop tail")


def _plan(n_samples, chunk, strict, downsampling, drop_short_rest):
    """The spans of one file: every full chunk, then the rest -- non-strict: the rest on its own; strict: the LAST
    `chunk` samples, of which only the frames the rest covers are kept (a rest shorter than one frame keeps, like the
    reference's `features[:, -0:]`, all of them).  drop_short_rest (the batched reader, :411): such a rest is ignored."""
    n_full = n_samples // chunk
    spans = [_Span(i * chunk, (i + 1) * chunk, 0) for i in range(n_full)]
    rest = n_samples - n_full * chunk
    if rest == 0 or (drop_short_rest and rest < downsampling):
        return spans
    if not strict:
        spans.append(_Span(n_full * chunk, n_samples, 0))
    else:
        # (a file shorter than one chunk: the slice seq[-chunk:] is then the whole file, as in the reference)
        spans.append(_Span(max(0, n_samples - chunk), n_samples, rest // downsampling))
    return spans


def _extract(featureMaker, seq, spans, group, seqNorm):
    """Run the spans, `group` equal-length ones per model call, in file order; [1, frames, dim] on the host."""
    wave = seq.reshape(-1).to(_model_device(featureMaker))             # the file goes to the device once
    pieces = []
    i = 0
    with torch.no_grad():
        while i < len(spans):
            length = spans[i].stop - spans[i].start
            j = i + 1
            while j < len(spans) and j - i < group and spans[j].stop - spans[j].start == length and not spans[j].tail \
                    and not spans[i].tail:
                j += 1
            batch = torch.stack([wave[s.start:s.stop] for s in spans[i:j]]).unsqueeze(1)      # [n, 1, length]
            feats = featureMaker((batch, None))
            for row, span in zip(feats, spans[i:j]):
                row = row.unsqueeze(0)
                if seqNorm:
                    row = seqNormalization(row)
                pieces.append((row[:, -span.tail:] if span.tail else row).cpu())
            i = j
    return torch.cat(pieces, dim=1)


def _load(seqPath):
    return seqPath if torch.is_tensor(seqPath) else audio.load(seqPath)[0]


def buildFeature(featureMaker, seqPath, strict=False, maxSizeSeq=64000, seqNorm=False):
    """Features [1, frames, dim] of one audio file (or [channels, samples] tensor), one chunk of maxSizeSeq samples per
    model call, in order -- a model built with keepHidden carries its recurrent state from chunk to chunk (:323-367)."""
    seq = _load(seqPath)
    spans = _plan(seq.size(1), maxSizeSeq, strict, featureMaker.getDownsamplingFactor(), drop_short_rest=False)
    return _extract(featureMaker, seq, spans, 1, seqNorm)


def buildFeature_batch(featureMaker, seqPath, strict=False, maxSizeSeq=8000, seqNorm=False, batch_size=8):
    """The same with the full chunks of maxSizeSeq samples fed batch_size at a time (:370-433): independent chunks, so
    a keepHidden model does NOT stream here (as in the reference).  A rest shorter than one frame is dropped."""
    seq = _load(seqPath)
    spans = _plan(seq.size(1), maxSizeSeq, strict, featureMaker.getDownsamplingFactor(), drop_short_rest=True)
    return _extract(featureMaker, seq, spans, batch_size, seqNorm)
