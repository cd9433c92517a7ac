"""Configuration mirror of the reference's lib/cfgs.py (module globals + the attribute dict
``c``).  Only what the pruning path reads is meaningful here; the other keys are kept so that
code written against the reference's config (train.py's auto-generated flags, cfgs.py:123-163)
keeps importing.  Reference: lib/cfgs.py:1-121."""


class _AttrDict(dict):
    """attribute-access dict (easydict is not a dependency)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


edict = _AttrDict

c = edict()
gpu = 1
dataset = "imagenet"
amd_vis = '0,1,2,3,4,5,6,7'      # devices the pruning workers may use (cfgs.py:5 caffe_vis analogue)
layer = False
gt_feats = False
_points_dict_name = "points_dict"
mp = 0
alpha = 1e-3                     # cfgs.py:18 -- carried from layer to layer (decompose.py:491, 627)


class Action:
    train = 'train'
    layer = 'layer'
    c3 = 'c3'
    combine = 'combine'


class kernels:
    dic = 'dic'
    pruning = 'pruning'


class solvers:
    sk = 'sklearn'               # cfgs.py:38-44; here 'sklearn' means "sklearn-exact arithmetic on HIP"
    lowparams = 'lowparams'
    gd = 'gd'
    keras = 'keras'
    tls = 'tls'
    lightning = 'lightning'


class pruning_options:           # cfgs.py:47-51
    prb = 0
    vgg = 3
    resnet = 4
    single = 10


class Data:
    lmdb = 'lmdb'
    pro = 'pro'


class Models:
    vgg = 'vgg'
    xception = 'xception'
    resnet = 'resnet'
    rescifar = 'rescifar'


class vgg:
    model = 'temp/vgg.prototxt'
    weights = 'temp/vgg.caffemodel'
    accname = 'accuracy@5'
    flop = 15346630656           # cfgs.py:66


c.dic = edict()
c.dic.option = pruning_options.prb
c.dic.layeralpha = 1
c.dic.debug = 0
c.dic.afterconv = False
c.dic.fitfc = 0
c.dic.keep = 3.                  # cfgs.py:74  "4x"
c.dic.rank_tol = .1              # cfgs.py:75
c.dic.prepooling = 1
c.dic.alter = 0
c.dic.vh = 1
c.res = edict()
c.res.short = 0
c.res.bn = 1
c.Action = Action.train
c.mp = True
c.kernelname = 'dic'
c.fc_ridge = 0                   # cfgs.py:99
c.ls = 'linear'                  # cfgs.py:100
c.nonlinear_fc = 0
c.nofc = 0
c.splitconvrelu = True
c.nBatches = 500                 # cfgs.py:104
c.ntest = 0
c.nBatches_fc = c.nBatches * 10
c.frozen = 0
c.nPointsPerLayer = 10           # cfgs.py:108
c.fc_reg = True
c.autodet = False                # cfgs.py:110
c.solver = solvers.sk            # cfgs.py:111
c.shm = '/tmp'
c.log = 'logs/'
c.model = ''
# knobs that only exist in this implementation
c.cd_mode = 'device'             # 'device': one foreign call per dictionary() (cp_prune_layer, alpha search in one
                                 # launch); 'steps': same search via the individual entry points; 'host': one launch per fit
# rounding variants of the coordinate update (all reproduce every reference golden mask and per-fit
# (nnz, n_iter) log; each is bit-identical to the matching mode of the CPU oracle).  The drop-in default is sklearn's
# own operation sequence (both 0); bench.py and the batched engines opt into the faster forms (-80 cycles per step):
c.cd_reciprocal = 0              # 1: multiply by 1/(Qii+l2) instead of dividing (<= 1 ulp per step)
c.cd_delta = 0                   # 1: one axpy with (w_new - w_old) instead of sklearn's two


def set_nBatches(n):
    c.nBatches = n
    c.nBatches_fc = c.nBatches
