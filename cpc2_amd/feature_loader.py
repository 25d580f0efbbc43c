"""Checkpoint I/O and forward-only feature extraction on the MI355X kernels.

Mirrors /root/reference/cpc/feature_loader.py: FeatureModule (:15-54), CPCModule (:57-82), getCheckpointData
(:176-199), loadModel (:238-283, single-checkpoint case), get_module (:286-293), save_checkpoint (:296-304),
seqNormalization (:316-320), buildFeature (:323-367).  Checkpoints use the reference's layout
{"gEncoder", "cpcCriterion", "optimizer", "best"} and key names, so files written by either side load in the other.
"""
import argparse
import json
import os

import torch

from . import audio
from .model import CPCModel
from .train import getAR, getEncoder


class FeatureModule(torch.nn.Module):
    """feature_loader.py:15-54 (without the sklearn CCA projection)."""

    def __init__(self, featureMaker, get_encoded, collapse=False, cca_projection=None):
        super(FeatureModule, self).__init__()
        if cca_projection:
            raise NotImplementedError("cca_projection is not on the MI355X path")
        self.get_encoded = get_encoded
        self.featureMaker = featureMaker
        self.collapse = collapse

    @property
    def out_feature_dim(self):
        if self.get_encoded:
            return self.featureMaker.gEncoder.getDimOutput()
        return self.featureMaker.gAR.getDimOutput()

    def getDownsamplingFactor(self):
        return self.featureMaker.gEncoder.DOWNSAMPLING

    def forward(self, data):
        batchAudio, label = data
        if len(batchAudio.size()) == 4:
            batchAudio = batchAudio[:, 0]
        device = next(self.featureMaker.parameters()).device
        cFeature, encoded, _ = self.featureMaker(batchAudio.to(device), label)
        if self.get_encoded:
            cFeature = encoded
        if self.collapse:
            cFeature = cFeature.contiguous().view(-1, cFeature.size(2))
        return cFeature


def get_module(i_module):
    if isinstance(i_module, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        return get_module(i_module.module)
    if isinstance(i_module, FeatureModule):
        return get_module(i_module.featureMaker)
    return i_module


def save_checkpoint(model_state, criterion_state, optimizer_state, best_state, path_checkpoint):
    torch.save({"gEncoder": model_state, "cpcCriterion": criterion_state, "optimizer": optimizer_state,
                "best": best_state}, path_checkpoint)


def getCheckpointData(pathDir):
    """newest checkpoint_N.pt + logs + args of a run directory (feature_loader.py:176-199)."""
    if not os.path.isdir(pathDir):
        return None
    checkpoints = [x for x in os.listdir(pathDir)
                   if os.path.splitext(x)[1] == '.pt' and os.path.splitext(x[11:])[0].isdigit()]
    if not checkpoints:
        return None
    checkpoints.sort(key=lambda x: int(os.path.splitext(x[11:])[0]))
    data = os.path.join(pathDir, checkpoints[-1])
    with open(os.path.join(pathDir, 'checkpoint_logs.json'), 'rb') as f:
        logs = json.load(f)
    with open(os.path.join(pathDir, 'checkpoint_args.json'), 'rb') as f:
        args = argparse.Namespace(**json.load(f))
    return os.path.abspath(data), logs, args


def loadModel(pathCheckpoints, loadStateDict=True):
    """Build CPCModel(s) from checkpoint_args.json and load `gEncoder` (feature_loader.py:238-283; one
    checkpoint, no nested `load` chains, no ConcatenatedModel)."""
    if len(pathCheckpoints) != 1:
        raise NotImplementedError("ConcatenatedModel (several checkpoints) is not on the MI355X path")
    path = pathCheckpoints[0]
    _, _, locArgs = getCheckpointData(os.path.dirname(path))
    model = CPCModel(getEncoder(locArgs), getAR(locArgs))
    if loadStateDict:
        state_dict = torch.load(path, 'cpu')
        model.load_state_dict(state_dict["gEncoder"], strict=False)
    return model, locArgs.hiddenGar, locArgs.hiddenEncoder


def seqNormalization(out):
    # out.size() = Batch x Seq x Channels   (feature_loader.py:316-320)
    mean = out.mean(dim=1, keepdim=True)
    var = out.var(dim=1, keepdim=True)
    return (out - mean) / torch.sqrt(var + 1e-08)


def buildFeature(featureMaker, seqPath, strict=False, maxSizeSeq=64000, seqNorm=False):
    """features [1, Seq, Dim] of one audio file, computed in chunks of maxSizeSeq samples
    (feature_loader.py:323-367; seqPath may also be a [channels, samples] tensor)."""
    seq = seqPath if torch.is_tensor(seqPath) else audio.load(seqPath)[0]
    sizeSeq = seq.size(1)
    start, out = 0, []
    while start < sizeSeq:
        if strict and start + maxSizeSeq > sizeSeq:
            break
        end = min(sizeSeq, start + maxSizeSeq)
        subseq = seq[:, start:end].reshape(1, 1, -1)
        with torch.no_grad():
            features = featureMaker((subseq, None))
            if seqNorm:
                features = seqNormalization(features)
        out.append(features.detach().cpu())
        start += maxSizeSeq
    if strict and start < sizeSeq:
        subseq = seq[:, -maxSizeSeq:].reshape(1, 1, -1)
        with torch.no_grad():
            features = featureMaker((subseq, None))
            if seqNorm:
                features = seqNormalization(features)
        delta = (sizeSeq - start) // featureMaker.getDownsamplingFactor()
        out.append(features[:, -delta:].detach().cpu())
    return torch.cat(out, dim=1)
