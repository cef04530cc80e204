"""``mx.io`` subset: DataIter / DataBatch / DataDesc (lib/iterators/MNIteratorBase.py:6, :216-217)."""
import collections


class DataDesc(collections.namedtuple('DataDesc', ['name', 'shape'])):
    def __new__(cls, name, shape, dtype='float32', layout='NCHW'):
        ret = super(DataDesc, cls).__new__(cls, name, tuple(shape))
        ret.dtype, ret.layout = dtype, layout
        return ret


class DataBatch(object):
    def __init__(self, data, label=None, pad=None, index=None, bucket_key=None, provide_data=None, provide_label=None):
        self.data, self.label, self.pad, self.index = data, label, pad, index
        self.bucket_key, self.provide_data, self.provide_label = bucket_key, provide_data, provide_label


class DataIter(object):
    def __init__(self, batch_size=0):
        self.batch_size = batch_size

    def __iter__(self):
        return self

    def reset(self):
        pass

    def next(self):
        if self.iter_next():
            return DataBatch(data=self.getdata(), label=self.getlabel(), pad=self.getpad(), index=self.getindex())
        raise StopIteration

    def __next__(self):
        return self.next()

    def iter_next(self):
        return False

    def getdata(self):
        return None

    def getlabel(self):
        return None

    def getindex(self):
        return None

    def getpad(self):
        return 0
