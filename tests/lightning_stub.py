"""A minimal stand-in for ``pytorch_lightning.Trainer`` -- TEST INFRASTRUCTURE ONLY.

The reference's training entry (``deepspeech_pytorch/training.py:13-47``) builds the model and hands it to
``hydra.utils.instantiate(cfg.trainer).fit(model, data_loader)``; everything that touches the model after that is Lightning's
automatic-optimization loop.  Lightning is not in the image, so the tests drive the drop-in class through this restatement of the
loop's CONTRACT with a LightningModule (Lightning 1.7, the version ``requirements.txt:10`` asks for), in the order Lightning runs it:

    configure_optimizers()                                        -> [optimizer], [scheduler]           (model.py:273-297)
    per batch:   batch moved to the module's device (every tensor of the 4-tuple, as Lightning's transfer_batch_to_device does)
                 optimizer.zero_grad()
                 with autocast(float16) if precision == 16:  loss = training_step(batch, batch_idx)    (model.py:241-249)
                 GradScaler.scale(loss).backward()  if precision == 16  else  loss.backward()
                 GradScaler.unscale_(optimizer)     if precision == 16
                 clip_grad_norm_(parameters, gradient_clip_val)                                         (configs/an4.yaml:12: 400)
                 GradScaler.step(optimizer); update()  /  optimizer.step()
    per epoch:   scheduler.step()                                 (ExponentialLR, interval "epoch")
    checkpoint:  {"epoch", "global_step", "state_dict", "hyper_parameters", "optimizer_states", "lr_schedulers"}   (Lightning's
                 .ckpt layout: what utils.py:31 ``DeepSpeech.load_from_checkpoint`` and checkpoint.py read)
"""
import torch


class MiniTrainer:
    def __init__(self, max_epochs=1, precision=32, gradient_clip_val=0.0, limit_train_batches=None, init_scale=65536.0, dry_run=False,
                 **ignored):
        self.max_epochs, self.precision, self.gradient_clip_val = max_epochs, precision, gradient_clip_val
        self.limit_train_batches, self.init_scale, self.dry_run = limit_train_batches, init_scale, dry_run
        self.ignored_kwargs = dict(ignored)        # e.g. replace_sampler_ddp=False, callbacks=... (training.py:42-46)
        self.losses, self.skipped_steps, self.global_step, self.current_epoch = [], 0, 0, 0
        self.model = self.optimizer = self.scheduler = self.scaler = None
        self.fitted = []

    @property
    def amp(self):
        return str(self.precision) in ("16", "16-mixed")

    def setup(self, model):
        self.model = model
        opts, scheds = model.configure_optimizers()
        self.optimizer, self.scheduler = opts[0], scheds[0]
        if self.amp and self.scaler is None:
            self.scaler = torch.amp.GradScaler("cuda", init_scale=self.init_scale)

    def fit(self, model, datamodule):
        self.fitted.append((model, datamodule))
        self.setup(model)
        if self.dry_run:
            return
        model.train()
        dev = next(model.parameters()).device
        for epoch in range(self.current_epoch, self.max_epochs):
            for i, batch in enumerate(datamodule.train_dataloader()):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                self.train_batch(tuple(t.to(dev) if torch.is_tensor(t) else t for t in batch), i)
            self.scheduler.step()
            self.current_epoch = epoch + 1

    def train_batch(self, batch, batch_idx):
        model, opt, scaler = self.model, self.optimizer, self.scaler
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.amp):
            loss = model.training_step(batch, batch_idx)
        if scaler is not None:
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
        else:
            loss.backward()
        if self.gradient_clip_val:
            torch.nn.utils.clip_grad_norm_(model.parameters(), self.gradient_clip_val)
        if scaler is not None:
            before = scaler.get_scale()
            scaler.step(opt)
            scaler.update()
            if scaler.get_scale() < before:
                self.skipped_steps += 1
        else:
            opt.step()
        self.global_step += 1
        self.losses.append(float(loss.detach().item()))
        return loss

    def checkpoint(self):
        return {"epoch": self.current_epoch, "global_step": self.global_step, "state_dict": self.model.state_dict(),
                "hyper_parameters": dict(self.model.hparams), "optimizer_states": [self.optimizer.state_dict()],
                "lr_schedulers": [self.scheduler.state_dict()]}

    def save_checkpoint(self, path):
        torch.save(self.checkpoint(), path)

    def resume(self, model, path):
        """What ``trainer.fit(model, ckpt_path=...)`` restores: weights, optimizer state, scheduler, counters."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["state_dict"], strict=True)
        self.setup(model)
        self.optimizer.load_state_dict(ck["optimizer_states"][0])
        self.scheduler.load_state_dict(ck["lr_schedulers"][0])
        self.current_epoch, self.global_step = ck["epoch"], ck["global_step"]
