from .runner_pretrain import run_net as pretrain_run_net  # noqa: F401
from .runner_autoencoder import run_net as token_run_net  # noqa: F401
