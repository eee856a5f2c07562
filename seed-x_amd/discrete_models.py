"""``DiscreteModleIdentity`` stand-in (the reference's spelling; src/models/tokenizer/discrete_models.py:9-21): the
shipped configs put an identity between the ViT features and the de-tokenizer's resampler."""


class DiscreteModleIdentity:
    def __init__(self, *args, **kwargs):
        pass

    def to(self, *args, **kwargs):
        return self

    def eval(self):
        return self

    def forward(self, image_embeds, input_ids=None, text_attention_mask=None, text_embeds=None):
        return

    __call__ = forward

    def encode_image_embeds(self, image_embeds):
        return image_embeds
