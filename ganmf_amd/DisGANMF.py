"""DisGANMF recommender — host mirror of GANRec/DisGANMF.py:21-266 (binary MLP discriminator on
[float(uid) | profile], sigmoid cross-entropy + feature matching).  Same fit() keyword arguments
and hooks as the reference; all arithmetic runs in libganmf_hip.so."""
import numpy as np

from . import _lib as L
from .GANMF import GANMF, _TensorRef, glorot_uniform


class DisGANMF(GANMF):
    RECOMMENDER_NAME = 'DisGANMF'

    def __init__(self, URM_train, mode='user', seed=1234, verbose=False, is_experiment=False, device=0, devices=None,
                 dist_backend=None, world_size=None, score_contract=None):
        super(DisGANMF, self).__init__(URM_train, mode=mode, verbose=verbose, seed=seed, is_experiment=is_experiment,
                                       device=device, devices=devices, dist_backend=dist_backend, world_size=world_size,
                                       score_contract=score_contract)

    # tensor ids follow tf.get_collection order (DisGANMF.py:121): layer_l/kernel, layer_l/bias, D_output/{kernel,bias}
    def _d_names(self):
        names = []
        for l in range(self.d_layers):
            names += ['discriminator/layer_%d/kernel' % l, 'discriminator/layer_%d/bias' % l]
        return names + ['discriminator/D_output/kernel', 'discriminator/D_output/bias']

    def _get(self, tid):
        a = self.engine.get_tensor(tid)
        if tid < 100 and (tid % 2 == 1):
            return a.reshape(-1)           # biases are 1-D in the reference
        return a

    def _build_dis(self, num_factors, d_layers, d_nodes, d_hidden_act, batch_size, **hp):
        self.num_factors, self.d_layers, self.d_nodes, self.d_hidden_act = num_factors, d_layers, d_nodes, d_hidden_act
        if self.engine is not None:
            self.engine.close()
        self.engine = self._make_engine(num_factors, d_nodes, batch_size, model=L.MODEL_DISGANMF, d_layers=d_layers,
                                        d_act=d_hidden_act, mfma=self.mfma, **hp)
        self.engine.set_urm(self._URM_fit)
        self.engine.set_seen(self._URM_eval)
        self._reset_score_filter()      # score_contract (GANMF.__init__): the reference's own semantics unless "mf" was asked for
        self.params = {'D': [_TensorRef(i, n) for i, n in enumerate(self._d_names())],
                       'G': [_TensorRef(t, n) for t, n in self._G_TENSORS]}
        from .GANMF import _SessionShim
        self.sess = _SessionShim(self)

    def _init_weights_dis(self):
        if self.initial_weights is not None:
            w = self.initial_weights
        else:
            rng = np.random.RandomState(self.seed)
            w = {}
            fan_in = self.num_items + 1
            for l in range(self.d_layers):
                w["W%d" % l] = glorot_uniform(rng, (fan_in, self.d_nodes))
                w["b%d" % l] = np.zeros(self.d_nodes, np.float32)
                fan_in = self.d_nodes
            w["Wo"] = glorot_uniform(rng, (fan_in, 1))
            w["bo"] = np.zeros(1, np.float32)
            w["U"] = glorot_uniform(rng, (self.num_users, self.num_factors))
            w["V"] = glorot_uniform(rng, (self.num_items, self.num_factors))
        for l in range(self.d_layers):
            self.engine.set_tensor(2 * l, w["W%d" % l])
            self.engine.set_tensor(2 * l + 1, w["b%d" % l])
        self.engine.set_tensor(2 * self.d_layers, w["Wo"])
        self.engine.set_tensor(2 * self.d_layers + 1, w["bo"])
        self.engine.set_tensor(L.T_USER_EMB, w["U"])
        self.engine.set_tensor(L.T_ITEM_EMB, w["V"])

    # DisGANMF.py:83-85
    def fit(self, num_factors=10, d_layers=1, d_nodes=32, d_hidden_act='linear', epochs=300, batch_size=32, d_lr=1e-4,
            g_lr=1e-4, d_steps=1, g_steps=1, d_reg=0, g_reg=0, recon_coefficient=1e-2, allow_worse=None, freq=None,
            after=0, metrics=['MAP'], sample_every=None, validation_evaluator=None, validation_set=None):
        self.config = dict(locals())
        del self.config['self']
        if d_hidden_act not in L.ACT:
            raise ValueError("unknown activation %r" % (d_hidden_act,))
        self._build_dis(num_factors, d_layers, d_nodes, d_hidden_act, batch_size, d_lr=d_lr, g_lr=g_lr, d_reg=d_reg,
                        g_reg=g_reg, m=0.0, recon_coefficient=recon_coefficient)
        self._init_weights_dis()
        return self._epoch_loop(epochs, d_steps, g_steps, allow_worse, freq, after, metrics, sample_every,
                                validation_evaluator, validation_set)

    def autoencoder_codes(self):
        raise AttributeError("DisGANMF has no autoencoder")

    def saveModel(self, folder_path, file_name):
        """DisGANMF.py:264-266: the Saver bundle only (the reference writes no build_params here and has no
        loadModel); load_bundle() below is the build's inverse, given the architecture."""
        import os
        from .tf_bundle import write_bundle
        self._require_engine()
        os.makedirs(folder_path, exist_ok=True)
        write_bundle(os.path.join(folder_path, file_name),
                     {ref.name: self.sess.run(ref) for ref in self.params['D'] + self.params['G']})

    def load_bundle(self, folder_path, file_name, num_factors, d_layers, d_nodes, d_hidden_act='linear'):
        import os
        from .tf_bundle import read_bundle
        data = read_bundle(os.path.join(folder_path, file_name))
        self._build_dis(num_factors, d_layers, d_nodes, d_hidden_act, batch_size=32)
        for ref in self.params['D'] + self.params['G']:
            self.engine.set_tensor(ref.tid, data[ref.name])
