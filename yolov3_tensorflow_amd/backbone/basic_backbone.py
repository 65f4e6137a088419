"""Layer factories with the reference's names and argument meaning (backbone/basic_backbone.py:20-163), building the lazy
graph of ``yolov3_tensorflow_amd.engine`` instead of tf.keras layers.  ``input_x`` is an ``engine.Val``; every factory
returns a ``Val``.  Fusion onto the HIP kernels happens when a value is consumed (see engine.Graph.materialize)."""
from yolov3_tensorflow_amd import engine


class BasicBackbone(object):
    L2_CONV_DECAY = engine.L2_CONV_DECAY          # reference :11
    BN_L2_GAMMA_DECAY = engine.BN_L2_GAMMA_DECAY  # :12
    BN_MOMENTUM = engine.BN_MOMENTUM              # :13
    BN_EPSILON = engine.BN_EPSILON                # :14
    BATCH_SIZE_AXIS = 0
    ROW_AXIS = 1
    COL_AXIS = 2
    CHANNEL_AXIS = 3

    @classmethod
    def convolution(cls, input_x, filters, **conv_params):
        """reference :20-43 -- defaults 3x3, stride 1, 'same', no bias, he_normal, L2 5e-4"""
        conv_params.setdefault('kernel_size', (3, 3))
        conv_params.setdefault('strides', (1, 1))
        conv_params.setdefault('padding', 'same')
        conv_params.setdefault('use_bias', False)
        unknown = set(conv_params) - {'kernel_size', 'strides', 'padding', 'use_bias', 'name', 'kernel_initializer', 'filters'}
        if unknown:
            raise TypeError('unsupported Conv2D arguments: %s' % sorted(unknown))
        init = conv_params.get('kernel_initializer', 'he_normal')
        return input_x.g.convolution(input_x, conv_params.get('filters', filters), kernel_size=conv_params['kernel_size'],
                                     strides=conv_params['strides'], padding=conv_params['padding'],
                                     use_bias=conv_params['use_bias'], name=conv_params.get('name'), init=init)

    @classmethod
    def depthwise_conv(cls, input_x, **conv_params):
        """reference :45-66 -- defaults 3x3, stride 1, 'same', no bias, he_normal, L2 5e-4.  Runs as a one-group launch of the mixed
        depthwise kernel (MixNet18's block uses the four-group form, engine.Graph.mix_depthwise_conv_bn)"""
        conv_params.setdefault('kernel_size', (3, 3))
        conv_params.setdefault('strides', (1, 1))
        conv_params.setdefault('padding', 'same')
        conv_params.setdefault('use_bias', False)
        unknown = set(conv_params) - {'kernel_size', 'strides', 'padding', 'use_bias', 'depthwise_initializer', 'depthwise_regularizer'}
        if unknown:
            raise TypeError('unsupported DepthwiseConv2D arguments: %s' % sorted(unknown))
        if tuple(conv_params['strides']) != (1, 1) or conv_params['padding'] != 'same' or conv_params['use_bias']:
            raise NotImplementedError("DepthwiseConv2D on this path: stride 1, 'same', no bias (every reference call site, mixnet18.py:43)")
        return input_x.g.depthwise_conv(input_x, kernel_size=conv_params['kernel_size'])

    @classmethod
    def batch_normalization(cls, input_x):
        """reference :68-78"""
        return input_x.g.batch_normalization(input_x)

    @classmethod
    def activation(cls, input_x, activation='relu', **activation_params):
        """reference :80-90 -- every call site uses the default ReLU"""
        if activation != 'relu' or activation_params:
            raise NotImplementedError('only ReLU is used by the reference graphs')
        return input_x.g.activation(input_x)

    @classmethod
    def element_wise_add(cls, identity, residual, is_nin=False):
        """reference :102-125"""
        stride_width = int(round(identity.shape[cls.ROW_AXIS] / residual.shape[cls.ROW_AXIS]))
        stride_height = int(round(identity.shape[cls.COL_AXIS] / residual.shape[cls.COL_AXIS]))
        if is_nin:
            identity = cls.convolution(identity, filters=residual.shape[cls.CHANNEL_AXIS], kernel_size=(1, 1),
                                       strides=(stride_width, stride_height), padding='valid')
            identity = cls.batch_normalization(identity)
        return identity.g.add(identity, residual)

    @classmethod
    def conv_bn(cls, input_x, filters, **conv_params):
        """reference :127-138"""
        return cls.batch_normalization(cls.convolution(input_x, filters, **conv_params))

    @classmethod
    def depthwise_conv_bn(cls, input_x, **conv_params):
        """reference :140-150"""
        return cls.batch_normalization(cls.depthwise_conv(input_x, **conv_params))

    @classmethod
    def bn_activation(cls, input_x, activation='relu', **activation_params):
        """reference :152-163"""
        return cls.activation(cls.batch_normalization(input_x), activation=activation, **activation_params)

    @classmethod
    def max_pooling(cls, input_x):
        """keras.layers.MaxPooling2D(pool_size=(3, 3), strides=(2, 2), padding='same') (resnet18.py:60)"""
        return input_x.g.max_pool(input_x)
