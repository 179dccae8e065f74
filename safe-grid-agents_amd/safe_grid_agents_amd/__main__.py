"""`python -m safe_grid_agents_amd <core flags> <env> <agent> <agent flags>` -- the reference's `python main.py ...`
(reference main.py:13-58) without Ray Tune: same flag grammar and defaults, default seed `random.randrange(500)`,
default log dir runs/<env>/<agent>/<baseline|corrupt>/<seed>."""
import os
import random

from .trainer import prepare_parser, train, train_batched


def _spawn_ranks(n, argv):
    """--devices N without a launcher: start the N ranks (one per GPU) with torch.distributed.run as a CHILD process -- nothing
    in this process has touched the GPU yet -- and return its exit code."""
    import socket
    import subprocess
    import sys

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "safe_grid_agents_amd"] + list(argv))


def main(argv=None):
    import sys

    args = prepare_parser().parse_args(argv)
    if getattr(args, "devices", 1) > 1 and "WORLD_SIZE" not in os.environ:
        if getattr(args, "n_envs", 0) <= 0:
            raise SystemExit("--devices shards the batched trainer: give -N/--n-envs too")
        raise SystemExit(_spawn_ranks(args.devices, sys.argv[1:] if argv is None else argv))
    if args.seed is None:
        args.seed = random.randrange(500)
    if getattr(args, "disable_cuda", False):
        args.device = "cpu"  # only the DeepQ network can leave the GPU; the envs have no CPU path
    if args.log_dir is None:
        cheating = "baseline" if args.cheat else "corrupt"
        args.log_dir = os.path.join("runs", args.env_alias, args.agent_alias, cheating, str(args.seed))
    os.makedirs(args.log_dir, exist_ok=True)
    if getattr(args, "n_envs", 0) > 0:
        return train_batched(args)
    return train(args)


if __name__ == "__main__":
    main()
