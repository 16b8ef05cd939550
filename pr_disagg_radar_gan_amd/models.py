"""Model objects with the slice of the Keras surface the reference uses on its generator and
critic (``predict``, ``get_weights``/``set_weights``, ``save``, ``trainable``, ``inputs``), backed
by the HIP engine.  Reference: create_generator / create_discriminator
(gan_train_cwgangp_pixelnorm.py:272-357) and load_model + predict
(raindisagg_gan_pretrained.py:43-60)."""
import numpy as np

from . import weights as W

_ENGINES = {}


def get_engine(ndomain, min_batch, n_cond_channels=1):
    """One engine per (device, ndomain, n_cond_channels); regrown when a bigger batch is requested."""
    import torch
    from .engine import Engine, require_gpu
    require_gpu()
    key = (torch.cuda.current_device(), int(ndomain), int(n_cond_channels))
    eng = _ENGINES.get(key)
    if eng is None or eng.max_batch < min_batch:
        if eng is not None:
            eng.close()
        eng = Engine(ndomain=ndomain, max_batch=max(int(min_batch), 32), n_cond_channels=n_cond_channels)
        _ENGINES[key] = eng
    return eng


class _Model:
    kind = None

    def __init__(self, arrays, ndomain, n_cond_channels=1):
        self.ndomain = int(ndomain)
        self.n_cond_channels = nc = int(n_cond_channels)
        self.shapes = W.gen_param_shapes(ndomain, nc) if self.kind == "generator" else W.critic_param_shapes(ndomain, nc)
        self.trainable = True
        self._slab = None
        self.set_weights(arrays)

    # Keras-like weight access (model.get_weights() order)
    def get_weights(self):
        if self._slab is not None:
            self._arrays = W.unflatten(self._slab.cpu().numpy(), self.shapes)
        return [a.copy() for a in self._arrays]

    def set_weights(self, arrays):
        arrays = [np.asarray(a, np.float32) for a in arrays]
        if len(arrays) != len(self.shapes):
            raise ValueError(f"{self.kind}: expected {len(self.shapes)} weight tensors, got {len(arrays)}")
        for a, (n, s) in zip(arrays, self.shapes):      # bound by position and shape, never by name
            if tuple(a.shape) != tuple(s):
                raise ValueError(f"{self.kind}: weight {n} has shape {a.shape}, expected {s}")
        self._arrays = arrays
        self._slab = None
        self._version = 0

    def count_params(self):
        return W.param_count(self.shapes)

    def device_slab(self, engine):
        if self._slab is None:
            from .engine import new_version
            self._slab = engine.to_slab(self._arrays)
            self._version = new_version()        # this object wrote the slab: it can vouch for its content (weight-form cache)
        return self._slab

    def adopt_slab(self, slab):
        """Share the trainer's device slab (weights then track training without copies)."""
        self._slab = slab
        self._version = 0                        # written by the trainer: content version unknown here, forms rebuilt per call

    def save(self, path):
        """``model.save(path)`` (reference :520-521): Keras-layout .h5 when h5py is importable, .npz otherwise."""
        W.save_weights(path, self.get_weights(), self.shapes, self.kind)

    def save_weights(self, path):
        self.save(path)

    def load_weights(self, path):
        self.set_weights(W.load_weights(path))


class Generator(_Model):
    kind = "generator"

    @property
    def inputs(self):
        class _In:                                        # gen.inputs[0].shape[1] == latent_dim (reference P:47)
            def __init__(self, shape):
                self.shape = shape
        nd = self.ndomain
        return [_In((None, W.LATENT_DIM)), _In((None, nd, nd, self.n_cond_channels))]

    def predict(self, inputs, batch_size=None, verbose=0):
        """generator.predict([latent, cond]) -> float32 ndarray (n, 24, nd, nd, 1)."""
        import torch
        latent, cond = inputs
        latent = np.ascontiguousarray(latent, dtype=np.float32)
        cond = np.ascontiguousarray(cond, dtype=np.float32)
        n, nd, nc = latent.shape[0], self.ndomain, self.n_cond_channels
        if latent.shape != (n, W.LATENT_DIM) or cond.shape != (n, nd, nd, nc):
            raise ValueError(f"predict expects latent (n,{W.LATENT_DIM}) and cond (n,{nd},{nd},{nc}); got {latent.shape}, {cond.shape}")
        chunk = int(batch_size or min(n, 1024))
        eng = get_engine(nd, chunk, nc)
        slab = self.device_slab(eng)
        out = np.empty((n, W.NHOURS, nd, nd, 1), np.float32)
        for i in range(0, n, chunk):
            z = torch.from_numpy(latent[i:i + chunk]).to(eng.device)
            c = torch.from_numpy(cond[i:i + chunk]).to(eng.device)
            res = eng.gen_forward(slab, z, c, gen_version=self._version)
            eng.check_numerics()                 # tf.debugging.check_numerics in the generator graph (reference T:349-350)
            out[i:i + chunk] = res.cpu().numpy()
        return out


class Critic(_Model):
    kind = "critic"

    def predict(self, inputs, batch_size=None, verbose=0):
        """critic.predict([sample, cond]) -> (n, 1); dropout off, as Keras ``predict`` does."""
        import torch
        sample, cond = inputs
        sample = np.ascontiguousarray(sample, dtype=np.float32)
        cond = np.ascontiguousarray(cond, dtype=np.float32)
        n, nd = sample.shape[0], self.ndomain
        chunk = int(batch_size or min(n, 1024))
        eng = get_engine(nd, max(1, (chunk + 2) // 3), self.n_cond_channels)
        chunk = min(chunk, 3 * eng.max_batch)
        slab = self.device_slab(eng)
        out = np.empty((n, 1), np.float32)
        for i in range(0, n, chunk):
            x = torch.from_numpy(sample[i:i + chunk]).to(eng.device)
            c = torch.from_numpy(cond[i:i + chunk]).to(eng.device)
            out[i:i + chunk] = eng.critic_forward(slab, x, c, seed=0, critic_version=self._version).cpu().numpy()
        return out


def load_generator(path):
    """tf.keras.models.load_model(generator_file, compile=False, custom_objects=...) (reference P:43-45)."""
    arrays = W.load_weights(path)
    return Generator(arrays, *W.infer_config_from_gen(arrays))
