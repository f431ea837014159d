from .runner_pretrain import run_net as pretrain_run_net  # noqa: F401
