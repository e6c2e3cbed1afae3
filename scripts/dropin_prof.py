import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.gan import Discriminator, Generator
torch.manual_seed(0)
BATCH = 64
generator, critic = Generator(), Discriminator()
critic.use_sigmoid = False
g_opt = torch.optim.RMSprop(generator.parameters(), lr=0.00005)
c_opt = torch.optim.RMSprop(critic.parameters(), lr=0.00005)
gen = torch.Generator().manual_seed(1000)
resident = [(torch.rand(BATCH, 32, 32, 32, generator=gen) * 2 - 1).cuda() for _ in range(5)]
def unit():
    for batch_index, batch in enumerate(resident):
        generator.zero_grad(); critic.zero_grad()
        fake_sample = generator.generate(sample_size=batch.shape[0]).detach()
        fake_out = critic(fake_sample); valid_out = critic(batch)
        critic_loss = torch.mean(fake_out) - torch.mean(valid_out)
        critic_loss.backward(); c_opt.step(); critic.clip_weights(0.01)
        if batch_index % 5 == 0:
            generator.zero_grad(); critic.zero_grad()
            fake_out = critic(generator.generate(sample_size=BATCH))
            generator_loss = -torch.mean(fake_out)
            generator_loss.backward(); g_opt.step()
            generator_loss.item()
for _ in range(3): unit()
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(5): unit()
torch.cuda.synchronize()
print("ms per unit", (time.perf_counter()-t0)/5*1e3)
