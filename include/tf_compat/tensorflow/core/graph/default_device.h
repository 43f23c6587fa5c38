// Shadow of <tensorflow/core/graph/default_device.h> for the HM drop-in: everything the reference needs from TensorFlow lives in
// pnn_tf_compat.h (see its header comment).
#include "pnn_tf_compat.h"   // needs -I<repo>/include next to -I<repo>/include/tf_compat
