"""`python -m safe_grid_agents_amd <core flags> <env> <agent> <agent flags>` -- the reference's `python main.py ...`
(reference main.py:13-58) without Ray Tune: same flag grammar and defaults, default seed `random.randrange(500)`,
default log dir runs/<env>/<agent>/<baseline|corrupt>/<seed>."""
import os
import random

from .trainer import prepare_parser, train, train_batched


def main(argv=None):
    args = prepare_parser().parse_args(argv)
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
