"""SURVEY.md §8 b6 / b2 in the build container (the reference tree is not shipped to the GPU box, so these run on CPU):

* the UNMODIFIED reference entry path -- concern/config.py compiling experiments/recognition/crnn.yaml, experiment.py
  constructing the Experiment (Structure / Builder / TrainSettings / OptimizerScheduler / Logger), data/data_loader.py
  collating a batch, trainer.py's Trainer.init_model + Trainer.train_step -- runs under megreader_amd.dropin.install()
  and reaches the HIP operator: on this GPU-less box the first kernel call raises the package's NotImplementedError
  ("no CPU fallback"), which is the proof that train_step -> SequenceRecognitionModel.forward -> BasicModel ->
  megreader_amd.backbones.crnn is wired.  (tests/test_crnn_gpu.py runs the same modules on the GPU.)
* level="extension": the reference's own ops/ctc_2d/ctc_loss_2d.py binds megreader_amd.ops.ctc_2d.ctc_2d_csrc under the
  name `ops.ctc_2d.ctc_2d_csrc`.
Both run in subprocesses: install() rewires sys.modules."""
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MEGREADER_REFERENCE", "/root/reference")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "experiments")), reason="reference tree not present")


def _run(code, cwd):
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + str(cwd), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=str(cwd), env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout


@needs_ref
def test_yaml_to_trainer_train_step_reaches_the_hip_op(tmp_path):
    # cwd-relative resources of the reference (concern/config.py:20-21, concern/charsets.py:68, concern/log.py:87-101)
    os.symlink(os.path.join(REF, "experiments"), tmp_path / "experiments")
    os.symlink(os.path.join(REF, "assets"), tmp_path / "assets")
    (tmp_path / "b6_synth.py").write_text(textwrap.dedent('''
        import torch
        from concern.config import Configurable, State
        class SyntheticRecognitionDataset(torch.utils.data.Dataset, Configurable):
            """stands in for FileDataset (the YAML's /data/text-spotter-data paths do not exist here)"""
            size = State(default=8)
            def __init__(self, cmd={}, **kwargs):
                self.load_all(**kwargs)
            def __len__(self):
                return self.size
            def __getitem__(self, i):
                g = torch.Generator().manual_seed(i)
                lab = torch.zeros(32, dtype=torch.int32)
                lab[:5] = torch.randint(2, 38, (5,), generator=g, dtype=torch.int32)
                return {'image': torch.rand(3, 32, 128, generator=g) - 0.5, 'label': lab,
                        'length': torch.tensor(5, dtype=torch.int32)}
    '''))
    out = _run('''
        import megreader_amd.dropin as dropin
        installed = dropin.install(%r)
        assert "crnn_backbone" in installed["backbones"] and "CRNNDecoder" in installed["decoders"]
        assert "CTCRepresenter" in installed["structure.representers"]
        assert "SequenceRecognitionMeasurer" in installed["structure.measurers"]
        import torch
        from concern.config import Configurable, Config
        from experiment import Structure, TrainSettings, ValidationSettings, Experiment   # tagged YAML classes (train.py:11)
        from trainer import Trainer
        conf = Config()
        ea = conf.compile(conf.load('experiments/recognition/crnn.yaml'))['Experiment']     # train.py:58-59
        ea['train']['data_loader'] = {'class': 'data.data_loader.DataLoader', 'batch_size': 4, 'num_workers': 0,
                                      'dataset': {'class': 'b6_synth.SyntheticRecognitionDataset', 'size': 8}}
        ea['validation'] = None
        ea['evaluation'] = None
        ea.update(cmd={'distributed': False, 'local_rank': 0, 'debug': False, 'validate': False, 'visualize': False,
                       'verbose': False, 'batch_size': 4, 'num_workers': 0})
        experiment = Configurable.construct_class_from_config(ea)                             # train.py:60
        trainer = Trainer(experiment)                                                         # train.py:64
        model = trainer.init_model()                                                          # trainer.py:38-41
        net = model.model.module if hasattr(model.model, "module") else model.model
        assert type(net.backbone).__module__ == "megreader_amd.backbones.crnn", type(net.backbone)
        assert type(net.decoder).__module__ == "megreader_amd.decoders.crnn", type(net.decoder)
        assert sum(p.numel() for p in model.parameters()) == 8332966                          # SURVEY.md §8b [probe]
        # evaluation side of the YAML (crnn.yaml:58-62): representer / measurer resolve to the GPU mirrors, built by the
        # reference's config system with the charset object it constructed from `^charset`
        rep, mea = experiment.structure.representer, experiment.structure.measurer
        assert type(rep).__module__ == "megreader_amd.structure.representers" and type(rep).__name__ == "CTCRepresenter"
        assert type(mea).__module__ == "megreader_amd.structure.measurers"
        assert type(rep.charset).__name__ == "EnglishCharset" and len(rep.charset) == 38
        optimizer = experiment.train.scheduler.create_optimizer(model.parameters())           # trainer.py:67-68
        assert type(optimizer).__name__ == "Adam"
        trainer.update_learning_rate(optimizer, 0, 0)
        assert abs(optimizer.param_groups[0]['lr'] - 1e-3) < 1e-12                            # crnn.yaml:82-89
        batch = next(iter(experiment.train.data_loader))
        assert tuple(batch['image'].shape) == (4, 3, 32, 128) and batch['label'].dtype == torch.int32
        model.train()
        try:
            trainer.train_step(model, optimizer, batch, epoch=0, step=0)                      # trainer.py:114-143
        except NotImplementedError as e:
            assert "megreader_amd" in str(e) and "CPU" in str(e), e
            print("REACHED_HIP_OP")
        else:
            raise SystemExit("train_step ran on a CPU box: a CPU fallback exists")
    ''' % REF, tmp_path)
    assert "REACHED_HIP_OP" in out


@needs_ref
def test_extension_level_binding_of_ctc_2d_csrc(tmp_path):
    out = _run('''
        import torch
        import megreader_amd.dropin as dropin
        dropin.install(%r, level="extension")
        import ops                                               # the REFERENCE's package (ops/__init__.py:1)
        import ops.ctc_2d.ctc_loss_2d as ref_fn                  # its own Function file, unchanged
        assert ref_fn.__file__.startswith(%r), ref_fn.__file__
        import megreader_amd.ops.ctc_2d.ctc_2d_csrc as ours
        assert ref_fn.ctc_2d_csrc is ours                        # `from . import ctc_2d_csrc` (ctc_loss_2d.py:3)
        assert ops.ctc_loss_2d is ref_fn.ctc_loss_2d
        for name in ("ctc2d_forward", "ctc2d_backward"):         # csrc/ctc2d.h:7-43
            assert callable(getattr(ours, name))
        lp = torch.zeros(4, 2, 1, 5).log_softmax(3)
        try:
            ops.ctc_loss_2d(lp, torch.zeros(1, 2, dtype=torch.long), torch.tensor([4]), torch.tensor([1]))
        except NotImplementedError:
            print("REFERENCE_CPU_CHECK")                         # ctc_loss_2d.py:12-13 raised, not ours
        try:
            ours.ctc2d_forward(lp, torch.zeros(1, 2, dtype=torch.long), torch.tensor([4]), torch.tensor([1]), 0, 1e-30)
        except RuntimeError as e:
            assert "CPU" in str(e)                               # AT_ERROR("Not implemented on the CPU"), ctc2d.h:20
            print("CSRC_CPU_ERROR")
    ''' % (REF, REF), tmp_path)
    assert "REFERENCE_CPU_CHECK" in out and "CSRC_CPU_ERROR" in out
