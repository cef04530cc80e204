"""``mx.metric`` subset (lib/train_utils/metric.py subclasses EvalMetric; main_train.py:113-133)."""


class EvalMetric(object):
    def __init__(self, name, num=None, output_names=None, label_names=None, **kwargs):
        self.name, self.num = name, num
        self.output_names, self.label_names = output_names, label_names
        self.reset()

    def update(self, labels, preds):
        raise NotImplementedError

    def update_dict(self, label, pred):
        self.update(list(label.values()), list(pred.values()))

    def reset(self):
        if self.num is None:
            self.num_inst, self.sum_metric = 0, 0.0
        else:
            self.num_inst, self.sum_metric = [0] * self.num, [0.0] * self.num

    def get(self):
        if self.num is None:
            return self.name, (float('nan') if self.num_inst == 0 else self.sum_metric / self.num_inst)
        names = ['%s_%d' % (self.name, i) for i in range(self.num)]
        vals = [x / y if y != 0 else float('nan') for x, y in zip(self.sum_metric, self.num_inst)]
        return names, vals

    def get_name_value(self):
        name, value = self.get()
        if not isinstance(name, list):
            name, value = [name], [value]
        return list(zip(name, value))


class CompositeEvalMetric(EvalMetric):
    def __init__(self, metrics=None, name='composite', **kwargs):
        self.metrics = list(metrics or [])
        super(CompositeEvalMetric, self).__init__(name)

    def add(self, metric):
        self.metrics.append(metric)

    def update(self, labels, preds):
        for m in self.metrics:
            m.update(labels, preds)

    def reset(self):
        for m in getattr(self, 'metrics', []):
            m.reset()

    def get(self):
        names, values = [], []
        for m in self.metrics:
            n, v = m.get()
            names.extend(n if isinstance(n, list) else [n])
            values.extend(v if isinstance(v, list) else [v])
        return names, values
