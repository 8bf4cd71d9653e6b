"""Command-line entry points with the flags of the reference's train_EEMFlow_HREM.py:139-156 / test_EEMFlow_HREM.py:124-140.

    python -m eemflow_amd.cli train [flags]      # train_EEMFlow_HREM.py: HREM meshflow training, checkpoint per epoch
    python -m eemflow_amd.cli test  [flags]      # test_EEMFlow_HREM.py: evaluation over every sequence of the HREM test split

Same flags and defaults as the reference scripts (`--lr --wd --train_iters --val_iters --batch_size --input_type --model_name
--start-epoch --num_workers --test_only --test_sequence ...`), the `lr{lr}_we{wd}` run folder, `config.json` / `train.log` /
`lasted_ckpt.pth.tar` in it, checkpoint loading with 'module.' stripping.  Added: `--data_root` (the reference hard-wires its own
repository path), `--save_root`, `--checkpoint`, `--config` (a JSON like the reference's config/a_meshflow.json; its training and
loader defaults are built in).  Dropped: visualisation, xlsx export, git metadata, nn.DataParallel (one process per GPU:
launch with torchrun for data parallelism; see eemflow_amd.parallel).  Only the models built here are accepted: EEMFlow (trained by
the fused step inside the library), `eraft` (train_EEMFlow_HREM.py:30-32) and `EEMFlow+` (both trained through the reference's own
statement sequence on the operator-level autograd route, train_mvsec.py:241-258).
"""
import argparse
import copy
import json
import os

import torch

# the fields of config/a_meshflow.json the scripts read (train_EEMFlow_HREM.py:25-68, train_mvsec.py:178-183)
DEFAULT_CONFIG = {
    "name": "mesh_flow",
    "train_img_size": [512, 960],
    "val_img_size": [720, 1280],
    "data_loader": {
        "train": {"args": {"batch_size": 6, "shuffle": True, "sequence_length": 1, "num_voxel_bins": 5, "eval_type": "dense",
                           "aug_params": {"crop_size": [512, 960], "min_scale": -0.1, "max_scale": 1.0, "do_flip": True}}},
        "test": {"args": {"batch_size": 1, "shuffle": False, "sequence_length": 1, "num_voxel_bins": 5, "align_to": "images",
                          "eval_type": "dense"}},
    },
    "train": {"lr": 1e-4, "wdecay": 5e-5, "epsilon": 1e-8, "num_steps": 1000000, "mixed_precision": True, "gamma": 0.8, "clip": 1.0},
}


# what the last train() of this process ended with (model, fused trainer, run folder, rank, world) - for callers that embed the CLI
# (tests compare the replicas of a data-parallel run through it); the scripts themselves only use the run folder
LAST_RUN = {}


def build_parser():
    p = argparse.ArgumentParser(prog="eemflow_amd.cli", description=__doc__.split("\n\n")[0])
    sub = p.add_subparsers(dest="command", required=True)

    def common(q, train):
        q.add_argument('-v', '--visualize', action='store_true', help='accepted for compatibility; visualisation is not built')
        q.add_argument('-n', '--num_workers', default=0, type=int, help='host threads that read and voxelize samples ahead (the reference: DataLoader worker processes); 0 = in the loop')
        q.add_argument('--train_iters', default=6000000 if train else 1000000, type=int, metavar='N', help='number of total iterations')
        q.add_argument('-se', '--start-epoch', action='store_true', help='restart from lasted_ckpt.pth.tar of the run folder')
        q.add_argument('-be', '--best_epe', default=1e5, type=float)
        q.add_argument('--val_iters', default=10000 if train else 3000, type=int, metavar='N', help='iterations per epoch (checkpoint interval)')
        q.add_argument('--lr', default=1e-5 if train else 1e-4, type=float, help='learning rate')
        q.add_argument('--wd', default=0 if train else 1e-5, type=float, help='weight decay')
        q.add_argument('--batch_size', '-bs', default=6 if train else 2, type=int, help='batch size in training')
        q.add_argument('--test_only', action='store_true')
        q.add_argument('--test_sequence', '-sq', default='indoor_flying2' if train else '', type=str)
        q.add_argument('--dense', action='store_true')
        q.add_argument('--density', '-d', default='ct0.05', type=str)
        q.add_argument('--model_name', '-model', default='EEMFlow', type=str)
        q.add_argument('--input_type', '-int', default='dt1', type=str)
        q.add_argument('--is_using_dynamic', '-dynamic', action='store_true')
        q.add_argument('--data_root', default=os.environ.get("EEMFLOW_DATA_ROOT", os.getcwd()), help='folder that holds dataset/HREM/...')
        q.add_argument('--save_root', default=os.getcwd(), help='where exp_HREM_meshflow/ (train) or HREM_testset/ (test) is created')
        q.add_argument('--checkpoint', default=None, help='test: checkpoint file (default <save_root>/checkpoints/EEMFlow_HREM_<input_type>.pth.tar)')
        q.add_argument('--config', default=None, help='JSON with the layout of config/a_meshflow.json (default: built-in copy of its used fields)')
        q.add_argument('--device', default='cuda:0')
        q.add_argument('--loader_threads', default=0, type=int, help='evaluation: host threads that read and voxelize samples ahead')
        q.add_argument('--frames_in_flight', default=1, type=int,
                       help='evaluation: samples kept in flight on as many model replicas / HIP streams (not in the reference; 4 suits one MI355X)')
        q.add_argument('--coalesce', default=1, type=int,
                       help='evaluation, EEMFlow: samples voxelized by one launch sequence and handed to one forward_many call (not in the '
                            'reference; 10 with --frames_in_flight 2 suits one MI355X); the volumes then stay raw and pconv1_1 normalises them')
    common(sub.add_parser('train', help='train_EEMFlow_HREM.py'), True)
    common(sub.add_parser('test', help='test_EEMFlow_HREM.py'), False)
    return p


def load_config(path):
    return json.load(open(path)) if path else copy.deepcopy(DEFAULT_CONFIG)


def build_model(name, config, training):
    if name == "EEMFlow":
        from .eemflow import EEMFlow
        # HREM's training target is the 16x16 mesh flow (HREM.py:254-255); the reference script builds the model without
        # out_mesh_size and its loss then meets a full-resolution prediction.  Training here predicts at mesh size
        # (EEMFlow.py:126-132), evaluation at full resolution against the upsampled mesh flow (HREM.py:264-267).
        return EEMFlow(config=config, n_first_channels=5, out_mesh_size=training)
    if name == "eraft":                                             # train_EEMFlow_HREM.py:30-32 / test_EEMFlow_HREM.py
        from .eraft import ERAFT
        split = 'train' if training else 'test'
        return ERAFT(config=config, n_first_channels=config['data_loader'][split]['args']['num_voxel_bins'])
    if name in ("EEMFlow+", "EEMFlow_cdc"):
        from .eemflow_plus import EEMFlow_cdc
        return EEMFlow_cdc(config=config, n_first_channels=5)
    raise SystemExit(f"model '{name}' is not built here (EEMFlow, eraft, EEMFlow+)")


def per_rank_batch(batch_size, world):
    """`--batch_size` is the GLOBAL batch, as in the reference, whose nn.DataParallel scatters it over the GPUs
    (train_EEMFlow_HREM.py:116-118): each of the `world` processes takes batch_size / world samples per step, so that the same flags
    (lr, train_iters, val_iters) train the same way on 1 and on 8 GPUs.  Equal shards are required: the gradient exchange is a
    mean of per-rank means."""
    if batch_size % world:
        raise SystemExit(f"--batch_size {batch_size} is not divisible by the {world} processes of this job")
    return batch_size // world


def train(args):
    from . import harness, parallel
    from .hrem import HREMEventFlow
    # one process per GPU under torchrun: every rank trains on its shard of the samples, the trainer all-reduces the flat gradient
    # (RCCL), rank 0 writes the logs and checkpoints
    rank, local_rank, world = parallel.init_distributed()
    if world > 1:
        args.device = "cuda:{}".format(parallel.local_device_index(local_rank))
        # this rank's host threads (the loader's workers inherit the mask) on the CPUs next to its GPU
        parallel.pin_host_threads_to_gpu_numa(parallel.local_device_index(local_rank))
    config = load_config(args.config)
    model = build_model(args.model_name, config, training=True)
    config["train"]["lr"] = args.lr                                                  # train_EEMFlow_HREM.py:56-59
    config["train"]["wdecay"] = args.wd
    config["train"]["num_steps"] = args.train_iters
    config['data_loader']['train']['args']['batch_size'] = args.batch_size
    config['name'] = "lr{:5f}_we{:5f}".format(args.lr, args.wd)
    save_path = os.path.join(args.save_root, "exp_HREM_meshflow/{}_{}".format(args.model_name, args.input_type), config['name'].lower())
    config["data_loader"]["train"]["args"].update({'type': 'train', 'event_interval': args.input_type})
    if rank == 0:                                                                    # one writer: rank 0 owns the run folder and the log
        os.makedirs(save_path, exist_ok=True)
        print('Storing output in folder {}'.format(save_path))
        json.dump(config, open(os.path.join(save_path, 'config.json'), 'w'), indent=4, sort_keys=False)
    start_epoch, start_iteration = 0, 0
    if args.start_epoch:
        start_epoch, start_iteration = harness.load_checkpoint(os.path.join(save_path, 'lasted_ckpt.pth.tar'), model, with_iteration=True)
    logger = harness.Logger(os.path.join(save_path, 'train.log') if rank == 0 else None, verbose=rank == 0)
    dev = torch.device(args.device)
    torch.cuda.set_device(dev)
    train_set = HREMEventFlow(args=config["data_loader"]["train"]["args"], train=True, root=args.data_root, device=dev)
    sampler = torch.utils.data.distributed.DistributedSampler(train_set, num_replicas=world, rank=rank, shuffle=True) if world > 1 else None
    if args.num_workers > 0:             # the reference's worker count: here host threads that read / inflate / voxelize samples ahead
        from .loader import ThreadedBatchLoader
        loader = ThreadedBatchLoader(train_set, per_rank_batch(args.batch_size, world), shuffle=sampler is None, sampler=sampler,
                                     threads=args.num_workers, drop_last=True)
    else:
        loader = torch.utils.data.DataLoader(train_set, batch_size=per_rank_batch(args.batch_size, world), shuffle=sampler is None,
                                             sampler=sampler, num_workers=0, drop_last=True)
    model = model.to(dev)
    if parallel.exchange_active():                                   # replicas start from rank 0's weights
        for prm in model.parameters():
            torch.distributed.broadcast(prm.data, src=0)
        if hasattr(model, "invalidate_weights"):
            model.invalidate_weights()                               # .data writes bypass the version counter the model watches
        else:
            from . import ops
            ops.invalidate_packed_weights()                          # the operator-level models cache packed weights by version too
    tcfg = config["train"]
    # The reference trainer sizes the padder with config['train_img_size'] (train_mvsec.py:67), the size its augmentor crops to.  The
    # HREM training samples here are the un-cropped frames, so the padder is sized from the first batch itself (image_size=None).
    # EEMFlow: the fused step inside the library; E-RAFT / EEMFlow+: the reference's statement sequence over the operator-level
    # autograd route (their data-parallel exchange is the flat-gradient all-reduce in TrainRaftEvents._train_iters_autograd)
    engine = "fused" if args.model_name == "EEMFlow" else "autograd"
    tr = harness.TrainRaftEvents(loader, None, lr=tcfg["lr"], wdecay=tcfg["wdecay"], epsilon=tcfg["epsilon"],
                                 num_steps=tcfg["num_steps"], clip=tcfg["clip"], gamma=tcfg["gamma"], logger=logger,
                                 start_iteration=start_iteration, engine=engine, mixed_precision=tcfg.get("mixed_precision", True))
    for epoch in range(start_epoch, max(args.train_iters // args.val_iters, 1)):
        if sampler is not None:
            sampler.set_epoch(epoch)
        model = tr.train_iters(model, start_epoch=epoch, val_iters=args.val_iters)
        if rank == 0:
            harness.save_checkpoint(os.path.join(save_path, 'lasted_ckpt.pth.tar'), model, epoch, trainer=tr.trainer, iteration=tr.iteration)
    parallel.barrier(dev)
    LAST_RUN.update(model=model, trainer=tr.trainer, save_path=save_path, rank=rank, world=world, engine=engine,
                    iteration=tr.trainer.iteration if tr.trainer is not None else tr.iteration)
    return save_path


def test(args):
    from . import harness
    from .hrem import HREMEventFlow
    config = load_config(args.config)
    model = build_model(args.model_name, config, training=False)
    ckpt = args.checkpoint or os.path.join(args.save_root, 'checkpoints', 'EEMFlow_HREM_{}.pth.tar'.format(args.input_type))
    start_epoch = harness.load_checkpoint(ckpt, model)
    save_path = os.path.join(args.save_root, "HREM_testset/{}_{}".format(args.model_name, args.input_type))
    os.makedirs(save_path, exist_ok=True)
    config["data_loader"]["test"]["args"].update({"event_interval": args.input_type})
    print('Storing output in folder {}'.format(save_path))
    json.dump(config, open(os.path.join(save_path, 'config.json'), 'w'), indent=4, sort_keys=False)
    logger = harness.Logger(os.path.join(save_path, 'test.log'))
    dev = torch.device(args.device)
    coalesce = args.coalesce if args.model_name == 'EEMFlow' else 1      # (forward_many is EEMFlow's)
    test_set = HREMEventFlow(args=config["data_loader"]["test"]["args"], train=False, root=args.data_root, device=dev,
                             deferred_norm=coalesce > 1)
    model = model.to(dev)
    sequences = [args.test_sequence] if args.test_sequence else list(test_set.nori_list.keys())
    ev = harness.TestRaftEvents(test_set, tuple(config["val_img_size"]), logger=logger)
    return ev.test_multi_sequence(model, start_epoch + 1, sequence_list=sequences, stride=1, frames_in_flight=args.frames_in_flight,
                                  loader_threads=args.loader_threads, coalesce=coalesce)


def main(argv=None):
    args = build_parser().parse_args(argv)
    return train(args) if args.command == 'train' else test(args)


if __name__ == '__main__':
    main()
