#!/usr/bin/env python3
"""Launcher of the sampling server with the reference launcher's command line (legion_server.py:72-85):

    python launch_server.py --dataset_path /data --dataset PA --gpu_number 8 --cache_memory 38000000000 --epoch 10

It writes the same one-line ``meta_config`` (legion_server.py:58-59:
``path batch V E F n_train n_valid n_test cache_bytes epochs partition_flag``) into the working directory and
starts ``csrc/legion <gpu_number> <cache_agg_mode> <fanouts>`` (the reference: ``./src/legion G mode``,
legion_server.py:69).  Unlike the reference, ``--nbrs_num`` is honoured (the reference hard-codes 25,10 in
Server.cu:68-69 and never forwards the flag).
"""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

# dataset facts the reference launcher carries (legion_server.py:6-53): directory, V, E, F, train/valid/test sizes
DATASETS = {
    "PR": ("products", 2449029, 123718280, 100, 196615, 39323, 2213091),
    "PA": ("paper100M", 111059956, 1615685872, 128, 11105995, 100000, 100000),
    "CO": ("com-friendster", 65608366, 1806067135, 256, 6560836, 100000, 100000),
    "UKS": ("ukunion", 133633040, 5507679822, 256, 13363304, 100000, 100000),
    "UKL": ("uk2014", 787801471, 47284178505, 128, 78780147, 100000, 100000),
    "CL": ("clueweb", 955207488, 42574107469, 128, 95520748, 100000, 100000),
}


def cache_agg_mode(gpu_number: int, usenvlink: int) -> int:
    """legion_server.py:61-68: pairs of GPUs form a clique when the fast interconnect is used."""
    return 1 if (usenvlink == 1 and gpu_number >= 2) else 0


def meta_line(args) -> str:
    name, V, E, F, n_train, n_valid, n_test = DATASETS[args.dataset]
    path = args.dataset_path + "/" + name + "/"
    return "{} {} {} {} {} {} {} {} {} {} {}".format(path, args.train_batch_size, V, E, F, n_train, n_valid, n_test,
                                                    args.cache_memory, args.epoch, 2 if args.seed_lists else 1 - args.usenvlink)


def main(argv=None):
    ap = argparse.ArgumentParser("Legion server (MI355X build).")
    ap.add_argument("--dataset_path", type=str, default="/home/atc-artifacts-user/datasets")
    ap.add_argument("--dataset", type=str, default="PA", choices=sorted(DATASETS))
    ap.add_argument("--train_batch_size", type=int, default=8000)
    ap.add_argument("--hops_num", type=int, default=2)
    ap.add_argument("--nbrs_num", type=str, default="25,10", help="fan-out per hop, e.g. 25,10,5")
    ap.add_argument("--gpu_number", type=int, default=1)
    ap.add_argument("--epoch", type=int, default=10)
    ap.add_argument("--cache_memory", type=int, default=38000000000)
    ap.add_argument("--usenvlink", type=int, default=1, help="1: cliques over the GPU interconnect (xGMI)")
    ap.add_argument("--seed_lists", action="store_true", help="extension (link prediction on several GPUs): every GPU g serves its own "
                    "pre-partitioned list trainingset_<G>_<g> verbatim (meta_config flag 2) instead of a split of `trainingset`")
    ap.add_argument("--dry_run", action="store_true", help="write meta_config and print the command only")
    args = ap.parse_args(argv)
    fan = [int(x) for x in args.nbrs_num.replace("[", "").replace("]", "").split(",") if x.strip()]
    with open("meta_config", "w") as f:
        f.write(meta_line(args))
    cmd = [os.path.join(HERE, "csrc", "legion"), str(args.gpu_number), str(cache_agg_mode(args.gpu_number, args.usenvlink)),
           ",".join(map(str, fan)), os.path.abspath("meta_config")]
    if args.dry_run:
        print(" ".join(cmd))
        return 0
    return subprocess.call(cmd)


if __name__ == "__main__":
    sys.exit(main())
